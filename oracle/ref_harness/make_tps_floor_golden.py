"""Reference-vs-itself floor of the two TPS paths across MKL code paths (build container only).

    python -m oracle.ref_harness.make_tps_floor_golden

Why: the fp32 ``torch.log`` of the TPS kernel term (core/udis_utils/torch_tps_transform.py:113-114,161; kornia_tps.py:45) is MKL VML's
vsLn (HA mode) on torch-CPU -- not SLEEF: the bits do not move with ATEN_CPU_CAPABILITY, they move with MKL_ENABLE_INSTRUCTIONS.  MKL
picks a different kernel per instruction set and the kernels do not agree: on 4e6 inputs of [0, 8) the AVX-512 kernel returns the
correctly rounded float on 99.97 %, the AVX2 kernel on 92.8 %, SSE4.2 on 92.2 %.  The (N+3)^2 solve amplifies those last-bit differences
of W, so the REFERENCE run on an AVX2 host differs from the reference run on an AVX-512 host -- same code, same inputs.  This script
measures that difference on the committed golden inputs (the parent process is the generator's own MKL path; children are started with
MKL_ENABLE_INSTRUCTIONS=AVX2 / SSE4_2) and stores it beside ``tps_T`` (the reference's solved coefficients, which `transformer` does not
return: restated here with the reference's own torch calls and CHECKED to reproduce the committed ``tps_out`` bit for bit).

Only data is written: tests/golden/tps_floor.npz.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import types

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")


def _reference_outputs():
    """What the reference computes here for the golden inputs: UDIS2 transformer output and the kornia-branch pipeline outputs."""
    from oracle import tps_pipeline as otp
    from oracle.ref_harness import make_tps_goldens as mk, stubs
    import contextlib
    import importlib
    import io
    mk.install_inference_stubs()
    G = np.load(os.path.join(OUT, "ops_small.npz"))
    U, src, tgt = (torch.from_numpy(G[k]) for k in ("tps_U", "tps_source", "tps_target"))
    res = {}
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        tps = importlib.import_module("core.udis_utils.torch_tps_transform")
        res["tps_out"] = tps.transformer(U, src, tgt, (24, 28)).numpy()
        # the reference's T, restated with the reference's own calls (:149-185)
        B, N, _ = src.shape
        p = torch.cat([torch.ones(B, N, 1).float(), src], 2)
        d2 = torch.sum(torch.square(p.reshape(B, -1, 1, 3) - p.reshape(B, 1, -1, 3)), 3)
        r = d2 * torch.log(d2 + 1e-6)
        W = torch.cat((torch.cat((p, r), 2), torch.cat((torch.zeros(B, 3, 3).float(), p.permute(0, 2, 1)), 2)), 1)
        Winv = torch.inverse(W.type(torch.float64))
        tp = torch.cat((tgt, torch.zeros(B, 3, 2)), 1)
        res["tps_T"] = torch.matmul(Winv, tp.type(torch.float64)).permute(0, 2, 1).type(torch.float32).numpy()
        res["tps_W_r"] = r.numpy()
        from core.inference import tps_pipline as ref_tp
        cfg = stubs.AttrDict(dict(grid_h=12, grid_w=12, pad_num=4, residual_flow_use_forward=False, flow_limit=-1, add_corner=False,
                                  get_pt_methods=["advanced_uniform_multi"], add_meshgrid=False, affine_scale=1.0, kernel_scale=1.0,
                                  use_boundary_limit=False, tps_method="kornia", output2_is_only_tps=True, do_avg_pooling=True))
        for name, seed, dims in (("a", 5, (200, 264, -21, -13, 236, 300)), ("b", 9, (160, 176, 0, -30, 211, 190))):
            ih, iw, wmin, hmin, oh, ow = dims
            case = otp.synthetic_case(seed, ih, iw, wmin, hmin, oh, ow)
            inputs = types.SimpleNamespace(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in case.items()})
            limit = types.SimpleNamespace(width_min=wmin, height_min=hmin, out_height=oh, out_width=ow)
            out = ref_tp.tps_H_warp(inputs, limit, cfg, inpaint_fn=None, is_plot=False)
            res[f"pipe_{name}_tps"] = out["tps_output"].numpy()
            res[f"pipe_{name}_blend"] = out["new_blend_image"].numpy()
    x = (np.random.default_rng(1).random(4_000_000, dtype=np.float32) * 8).astype(np.float32)
    lg = torch.log(torch.from_numpy(x)).numpy()
    res["log_vs_correctly_rounded_frac"] = np.array((lg != np.log(x.astype(np.float64)).astype(np.float32)).mean())
    res["log_probe"] = lg[:200_000]
    return res


def child(path):
    np.savez(path, **_reference_outputs())


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    here = _reference_outputs()
    G = np.load(os.path.join(OUT, "ops_small.npz"))
    assert np.array_equal(here["tps_out"], G["tps_out"]), "this process is not the MKL path the committed golden came from"
    out = {"tps_T": here["tps_T"], "generator_log_vs_cr_frac": here["log_vs_correctly_rounded_frac"]}
    # the restated T must be THE reference's T: it reproduces the committed tps_out through the oracle's grid / gather
    from oracle import cgeom, geom
    src, U = torch.from_numpy(G["tps_source"]), torch.from_numpy(G["tps_U"])
    B, N, _ = src.shape
    oh, ow = 24, 28
    xt = torch.from_numpy(cgeom.linspace(-1.0, 1.0, ow))[None, :].expand(oh, -1).reshape(1, 1, -1)
    yt = torch.from_numpy(cgeom.linspace(-1.0, 1.0, oh))[:, None].expand(-1, ow).reshape(1, 1, -1)
    dd = (xt - src[:, :, 0:1]) ** 2 + (yt - src[:, :, 1:2]) ** 2
    grid = torch.cat([torch.ones(B, 1, oh * ow), xt.expand(B, -1, -1), yt.expand(B, -1, -1), dd * torch.log(dd + 1e-6)], 1)
    Tg = torch.matmul(torch.from_numpy(here["tps_T"]), grid)
    assert np.array_equal(geom.tps_interpolate(U, Tg[:, 0], Tg[:, 1], (oh, ow)).numpy(), G["tps_out"])
    summary = {}
    for isa in ("AVX2", "SSE4_2"):
        tmp = os.path.join(OUT, f"_floor_{isa}.npz.tmp.npz")
        env = dict(os.environ, MKL_ENABLE_INSTRUCTIONS=isa)
        subprocess.check_call([sys.executable, "-m", "oracle.ref_harness.make_tps_floor_golden", "--child", tmp], env=env,
                              cwd=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
        other = np.load(tmp)
        os.remove(tmp)
        k = isa.lower()
        d = np.abs(other["tps_out"] - here["tps_out"])
        out[f"floor_{k}_tps_out_max"], out[f"floor_{k}_tps_out_p99"] = np.array(d.max()), np.array(np.percentile(d, 99))
        out[f"floor_{k}_tps_T_rel"] = np.array(np.abs(other["tps_T"] - here["tps_T"]).max() / np.abs(here["tps_T"]).max())
        out[f"floor_{k}_W_entries_differ"] = np.array(int((other["tps_W_r"] != here["tps_W_r"]).sum()))
        out[f"floor_{k}_log_differs_frac"] = np.array((other["log_probe"] != here["log_probe"]).mean())
        out[f"floor_{k}_log_vs_cr_frac"] = other["log_vs_correctly_rounded_frac"]
        for name in ("a", "b"):
            dp = np.abs(other[f"pipe_{name}_tps"] - here[f"pipe_{name}_tps"])
            db = np.abs(other[f"pipe_{name}_blend"].astype(np.int32) - here[f"pipe_{name}_blend"].astype(np.int32))
            out[f"floor_{k}_pipe_{name}_tps_p99"], out[f"floor_{k}_pipe_{name}_tps_max"] = np.array(np.percentile(dp, 99)), np.array(dp.max())
            out[f"floor_{k}_pipe_{name}_blend_differs_frac"] = np.array((db > 0).mean())
        summary[isa] = {kk[len(f"floor_{k}_"):]: float(v) for kk, v in out.items() if kk.startswith(f"floor_{k}_")}
    np.savez_compressed(os.path.join(OUT, "tps_floor.npz"), **out)
    print(json.dumps({"generator_log_vs_cr_frac": float(out["generator_log_vs_cr_frac"]), **summary}, indent=1))


if __name__ == "__main__":
    main()
