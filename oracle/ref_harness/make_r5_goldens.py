"""Round-5 goldens from the REFERENCE's own Python (build container only; nothing here runs on the GPU box).

    python -m oracle.ref_harness.make_r5_goldens

Writes tests/golden/e2e_r5_512.npz (data only):
  b2_*            `test_eval` of the reference on a BATCH OF TWO structured 512x512 pairs (seeds 7 and 11) with the damped weights:
                  the oracle-checked batch > 1 case of BASELINE configs[2] (the caller that batches is evaluate.py:34-43)
  b2_floor_*      the reference against itself on that batch, 8 vs 1 CPU threads
  demo{1,2}_floor_seeded_*   the reference against itself (8 vs 1 threads) on its two demo pairs with the seeded weights 1234: the floor under
                  every bound of test_end_to_end_reference_demo_pairs_512 (`test_eval` and `test_out`); the 8-thread run is checked to
                  reproduce the committed e2e_demo_512.npz first
  demo{1,2}_damped_*         `test_eval` of the reference on the demo pairs with the damped weights (the non-chaotic case) + its own floor
"""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import inputs, spec
from oracle.ref_harness import stubs
from oracle.ref_harness.make_e2e_goldens import eval_record, out_record
from oracle.ref_harness.make_goldens import OUT, packbits


def eval_floor(o8, o1, prefix):
    f8, f1 = o8["flow_predictions"][0], o1["flow_predictions"][0]
    d = (f8 - f1).abs().flatten()
    r = {prefix + "flow_max_px": float(d.max()), prefix + "flow_p99_px": float(d.kthvalue(int(0.99 * d.numel())).values),
         prefix + "occ_flips": int((o8["origin_occlusion_mask"] != o1["origin_occlusion_mask"]).sum()),
         prefix + "overlap_flips": int((o8["overlap"] != o1["overlap"]).sum()),
         prefix + "H_max": float((o8["H"] - o1["H"]).abs().max()),
         prefix + "H_rel": float((o8["H"] - o1["H"]).abs().max() / max(1.0, float(o8["H"].abs().max()))),
         prefix + "output_H_max": float((o8["output_H"] - o1["output_H"]).abs().max()),
         prefix + "output_H_p99": float(np.percentile((o8["output_H"] - o1["output_H"]).abs().numpy()[..., ::8, ::8], 99)),
         prefix + "final_max": float((o8["final_warp_output"] - o1["final_warp_output"]).abs().max())}
    return {k: np.array(v) for k, v in r.items()}


def out_floor(o8, o1, prefix):
    r = {prefix + "ints_equal": int(all(o8[k] == o1[k] for k in ("width_min", "height_min", "out_height", "out_width"))),
         prefix + "H_rel": float((o8["H"] - o1["H"]).abs().max() / max(1.0, float(o8["H"].abs().max())))}
    if r[prefix + "ints_equal"]:
        d = (o8["blend_image"].int() - o1["blend_image"].int()).abs()[..., ::2, ::2]
        r[prefix + "blend_gt2_frac"] = float((d > 2).float().mean())
        r[prefix + "blend_differs_frac"] = float((d > 0).float().mean())
        drf = (o8["residual_flow"] - o1["residual_flow"]).abs()[..., ::8, ::8]
        r[prefix + "residual_flow_p99_px"] = float(np.percentile(drf.numpy(), 99))
        for k in ("mask1", "mask2", "occlusion_mask", "origin_occlusion_mask", "warp_input2_mask"):
            r[prefix + k + "_flip_frac"] = float(((o8[k] >= 0.5) != (o1[k] >= 0.5)).float().mean())
    return {k: np.array(v) for k, v in r.items()}


def main():
    from PIL import Image
    torch.manual_seed(0)
    overlay = dict(test_not_use_combine_h_flow=True, use_forward=False, use_fb_consistency_mask=True, use_whole_resolution=False)
    rec = {}
    demos = {}
    for name in ("demo1", "demo2"):
        arrs = [np.asarray(Image.open(f"{stubs.REF_ROOT}/demo/{name}/input{i}.jpg").convert("RGB")).copy() for i in (1, 2)]
        demos[name] = tuple(torch.from_numpy(x).permute(2, 0, 1)[None].float() for x in arrs)

    def run(model, fn, threads):
        torch.set_num_threads(threads)
        with torch.no_grad():
            r = fn(model)
        torch.set_num_threads(8)
        return r

    # ---- damped weights: the batch of two + the demo pairs -------------------------------------------------------------------
    model, _ = stubs.build_reference(spec.damped_state_dict(1234), overlay=overlay)
    p0, p1 = inputs.structured_pair(512, 512, seed=7), inputs.structured_pair(512, 512, seed=11, shift=(5, -3))
    A, B = torch.cat([p0[0], p1[0]]), torch.cat([p0[1], p1[1]])
    rec["b2_seeds"] = np.array([7, 11]); rec["b2_shift1"] = np.array([5, -3])
    o8 = run(model, lambda m: m(A, B, type="test_eval"), 8)
    o1 = run(model, lambda m: m(A, B, type="test_eval"), 1)
    rec.update(eval_record(o8, "b2_", s=4))
    rec.update(eval_floor(o8, o1, "b2_floor_"))
    # sample 0 of the batch is the pair of e2e_eval_damped_512.npz: the reference's batch-2 forward against its own batch-1 forward
    g1 = np.load(os.path.join(OUT, "e2e_eval_damped_512.npz"))
    rec["b2_vs_b1_flow_max_px"] = np.array(float(np.abs(o8["flow_predictions"][0][0:1, :, ::4, ::4].numpy() - g1["flow_sub"]).max()))
    rec["b2_vs_b1_occ_flips"] = np.array(int(np.unpackbits(packbits(o8["origin_occlusion_mask"][0:1]) ^ g1["occ_bits"]).sum()))
    print("b2", {k: v.tolist() for k, v in rec.items() if "floor" in k or "vs_b1" in k}, flush=True)
    for name, (ta, tb) in demos.items():
        o8 = run(model, lambda m: m(ta, tb, type="test_eval"), 8)
        o1 = run(model, lambda m: m(ta, tb, type="test_eval"), 1)
        rec.update(eval_record(o8, name + "_damped_", s=8))
        rec.update(eval_floor(o8, o1, name + "_damped_floor_"))
        print(name, "damped", {k: v.tolist() for k, v in rec.items() if k.startswith(name + "_damped_floor")}, flush=True)

    # ---- seeded weights: floor of the committed demo goldens --------------------------------------------------------------------
    model, _ = stubs.build_reference(spec.seeded_state_dict(1234), overlay=overlay)
    gd = np.load(os.path.join(OUT, "e2e_demo_512.npz"))
    for name, (ta, tb) in demos.items():
        e8 = run(model, lambda m: m(ta, tb, type="test_eval"), 8)
        assert np.array_equal(e8["H"].numpy(), gd[name + "_eval_H"]) and np.array_equal(packbits(e8["origin_occlusion_mask"]), gd[name + "_eval_occ_bits"]), \
            "the 8-thread run does not reproduce the committed demo golden"
        e1 = run(model, lambda m: m(ta, tb, type="test_eval"), 1)
        rec.update(eval_floor(e8, e1, name + "_floor_seeded_eval_"))
        t8 = run(model, lambda m: m(ta, tb, type="test_out"), 8)
        t1 = run(model, lambda m: m(ta, tb, type="test_out"), 1)
        rec.update(out_floor(t8, t1, name + "_floor_seeded_out_"))
        print(name, "seeded", {k: v.tolist() for k, v in rec.items() if k.startswith(name + "_floor_seeded")}, flush=True)
    np.savez_compressed(os.path.join(OUT, "e2e_r5_512.npz"), **rec)
    print("e2e_r5_512.npz", os.path.getsize(os.path.join(OUT, "e2e_r5_512.npz")))


if __name__ == "__main__":
    main()
