"""Second reference-vs-itself floor for tests/golden/e2e_r5_512.npz: the reference under another MKL code path (build container only).

    python -m oracle.ref_harness.make_r5_isa_floor

`make_r5_goldens.py` recorded the 8-vs-1-thread floor.  A thread count only changes how MKL splits some GEMMs; another HOST changes every
MKL kernel (sgemm blocking, the vector math of GELU / softmax / exp): the reference run with MKL_ENABLE_INSTRUCTIONS=AVX2 (what an EPYC or
an older Xeon would execute) against the reference run here (AVX-512), same code, same inputs, 8 threads.  This build replaces every one
of those kernels, so this is the floor its own differences are comparable with.  Adds `*_floor_avx2_*` keys to e2e_r5_512.npz.
"""
from __future__ import annotations

import os
import subprocess
import sys

import numpy as np
import torch

from oracle import inputs, spec
from oracle.ref_harness import stubs
from oracle.ref_harness.make_goldens import OUT
from oracle.ref_harness.make_r5_goldens import eval_floor

CASES = (("demo1", "seeded"), ("demo2", "seeded"), ("demo1", "damped"), ("demo2", "damped"), ("b2", "damped"), ("struct", "damped"), ("struct", "seeded"))


def run_cases():
    from PIL import Image
    torch.manual_seed(0)
    torch.set_num_threads(8)
    overlay = dict(test_not_use_combine_h_flow=True, use_forward=False, use_fb_consistency_mask=True, use_whole_resolution=False)
    res = {}
    models = {}
    for name, wk in CASES:
        if wk not in models:
            sd = spec.seeded_state_dict(1234) if wk == "seeded" else spec.damped_state_dict(1234)
            models[wk] = stubs.build_reference(sd, overlay=overlay)[0]
        if name.startswith("demo"):
            arrs = [np.asarray(Image.open(f"{stubs.REF_ROOT}/demo/{name}/input{i}.jpg").convert("RGB")).copy() for i in (1, 2)]
            ta, tb = (torch.from_numpy(x).permute(2, 0, 1)[None].float() for x in arrs)
        elif name == "b2":
            p0, p1 = inputs.structured_pair(512, 512, seed=7), inputs.structured_pair(512, 512, seed=11, shift=(5, -3))
            ta, tb = torch.cat([p0[0], p1[0]]), torch.cat([p0[1], p1[1]])
        else:
            ta, tb = inputs.structured_pair(512, 512, seed=7)
        with torch.no_grad():
            o = models[wk](ta, tb, type="test_eval")
        for k in ("H", "output_H", "final_warp_output", "overlap", "origin_occlusion_mask"):
            res[f"{name}_{wk}_{k}"] = o[k].numpy()
        res[f"{name}_{wk}_flow"] = o["flow_predictions"][0].numpy()
        print(name, wk, "done", flush=True)
    return res


def as_out(res, name, wk):
    g = lambda k: torch.from_numpy(res[f"{name}_{wk}_{k}"])          # noqa: E731
    return {"H": g("H"), "output_H": g("output_H"), "final_warp_output": g("final_warp_output"), "overlap": g("overlap"),
            "origin_occlusion_mask": g("origin_occlusion_mask"), "flow_predictions": [g("flow")]}


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        np.savez(sys.argv[2], **run_cases())
        return
    tmp = os.path.join(OUT, "_r5_isa_child.npz.tmp.npz")
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
    subprocess.check_call([sys.executable, "-m", "oracle.ref_harness.make_r5_isa_floor", "--child", tmp],
                          env=dict(os.environ, MKL_ENABLE_INSTRUCTIONS="AVX2"), cwd=root)
    other = dict(np.load(tmp))
    os.remove(tmp)
    here = run_cases()
    path = os.path.join(OUT, "e2e_r5_512.npz")
    rec = dict(np.load(path))
    for name, wk in CASES:
        prefix = f"{name}_{wk}_floor_avx2_"
        rec.update(eval_floor(as_out(here, name, wk), as_out(other, name, wk), prefix))
        print(prefix, {k[len(prefix):]: v.tolist() for k, v in rec.items() if k.startswith(prefix)}, flush=True)
    np.savez_compressed(path, **rec)
    print(os.path.getsize(path))


if __name__ == "__main__":
    main()
