"""Round-4 end-to-end goldens from the REFERENCE's own Python (build container only; nothing here runs on the GPU box).

    python -m oracle.ref_harness.make_e2e_goldens

Writes (data only: decoded inputs, expected outputs, checksums):
  tests/golden/e2e_demo_512.npz        `test_eval` and `test_out` of the reference on its two shipped photo pairs
                                       (demo/demo1, demo/demo2: 512x512 JPEGs, native size), seeded weights 1234
  tests/golden/e2e_eval_damped_512.npz `test_eval` on the structured 512x512 pair with ``spec.damped_state_dict`` (loop gain of the
                                       refinement < 1), plus the reference's own floor on that case: the same run with 1 instead
                                       of 8 CPU threads (flow difference, occlusion flips)
"""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import inputs, spec
from oracle.ref_harness import stubs
from oracle.ref_harness.make_goldens import OUT, checksum, packbits, sub


def eval_record(o, prefix, s=8):
    f = o["flow_predictions"][0]
    return {prefix + "H": o["H"].numpy(), prefix + "flow_sub": sub(f, s), prefix + "flow_cs": checksum(f),
            prefix + "output_H_sub": sub(o["output_H"], s), prefix + "output_H_cs": checksum(o["output_H"]),
            prefix + "output_H_inv_cs": checksum(o["output_H_inv"]),
            prefix + "final_sub": sub(o["final_warp_output"], s), prefix + "final_cs": checksum(o["final_warp_output"]),
            prefix + "overlap_bits": packbits(o["overlap"]), prefix + "occ_bits": packbits(o["origin_occlusion_mask"])}


def out_record(o, prefix):
    return {prefix + "blend_sub": o["blend_image"][..., ::2, ::2].contiguous().numpy(), prefix + "blend_cs": checksum(o["blend_image"]),
            prefix + "H": o["H"].numpy(), prefix + "I_mat": o["I_mat"].numpy(),
            prefix + "ints": np.array([o["width_min"], o["height_min"], o["out_height"], o["out_width"]]),
            prefix + "residual_flow_sub": sub(o["residual_flow"], 8), prefix + "residual_flow_cs": checksum(o["residual_flow"]),
            prefix + "mask1_bits": packbits(o["mask1"]), prefix + "mask2_bits": packbits(o["mask2"]),
            prefix + "occ_bits": packbits(o["occlusion_mask"]), prefix + "origin_occ_bits": packbits(o["origin_occlusion_mask"]),
            prefix + "warp_mask_bits": packbits(o["warp_input2_mask"]),
            prefix + "output2_cs": checksum(o["output2"]), prefix + "final_warp_cs": checksum(o["final_warp"]),
            prefix + "H_warp_cs": checksum(o["H_warp"]), prefix + "final_warp_sub": sub(o["final_warp"], 8)}


def main():
    from PIL import Image
    torch.manual_seed(0)
    torch.set_num_threads(8)
    overlay = dict(test_not_use_combine_h_flow=True, use_forward=False, use_fb_consistency_mask=True, use_whole_resolution=False)
    model, _ = stubs.build_reference(spec.seeded_state_dict(1234), overlay=overlay)
    rec = {}
    for name in ("demo1", "demo2"):
        arrs = [np.asarray(Image.open(f"{stubs.REF_ROOT}/demo/{name}/input{i}.jpg").convert("RGB")).copy() for i in (1, 2)]
        assert arrs[0].shape == (512, 512, 3)
        ta, tb = (torch.from_numpy(x).permute(2, 0, 1)[None].float() for x in arrs)
        rec[name + "_input1"], rec[name + "_input2"] = arrs
        with torch.no_grad():
            rec.update(eval_record(model(ta, tb, type="test_eval"), name + "_eval_"))
            rec.update(out_record(model(ta, tb, type="test_out"), name + "_out_"))
        print(name, "done", flush=True)
    np.savez_compressed(os.path.join(OUT, "e2e_demo_512.npz"), **rec)

    # ---- damped end-to-end case + the reference's own thread-count floor on it ---------------------------------------------
    model, _ = stubs.build_reference(spec.damped_state_dict(1234), overlay=overlay)
    a, b = inputs.structured_pair(512, 512, seed=7)
    with torch.no_grad():
        o8 = model(a, b, type="test_eval")
        torch.set_num_threads(1)
        o1 = model(a, b, type="test_eval")
        torch.set_num_threads(8)
    rec = eval_record(o8, "", s=4)
    f8, f1 = o8["flow_predictions"][0], o1["flow_predictions"][0]
    d = (f8 - f1).abs().flatten()
    rec["flow_scale"] = np.array(spec.DAMPED_FLOW_SCALE)
    rec["ref_floor_flow_max_px"] = np.array(float(d.max()))
    rec["ref_floor_flow_p99_px"] = np.array(float(d.kthvalue(int(0.99 * d.numel())).values))
    rec["ref_floor_occ_flips"] = np.array(int((o8["origin_occlusion_mask"] != o1["origin_occlusion_mask"]).sum()))
    rec["ref_floor_H_max"] = np.array(float((o8["H"] - o1["H"]).abs().max()))
    rec["ref_floor_final_max"] = np.array(float((o8["final_warp_output"] - o1["final_warp_output"]).abs().max()))
    rec["ref_floor_output_H_max"] = np.array(float((o8["output_H"] - o1["output_H"]).abs().max()))
    rec["flow_absmax"] = np.array(float(f8.abs().max()))
    rec["occluded_px"] = np.array(int((o8["origin_occlusion_mask"] == 0).sum()))
    print({k: v.tolist() for k, v in rec.items() if k.startswith("ref_floor") or k in ("flow_absmax", "occluded_px")})
    np.savez_compressed(os.path.join(OUT, "e2e_eval_damped_512.npz"), **rec)
    for f in ("e2e_demo_512.npz", "e2e_eval_damped_512.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
