"""Round-6 golden from the REFERENCE's own Python (build container only; nothing here runs on the GPU box).

    python -m oracle.ref_harness.make_r6_goldens

Writes tests/golden/tps_photo_512.npz (data only): the UDIS2 TPS `transformer` (core/udis_utils/torch_tps_transform.py:7-190) applied to IMAGE
CONTENT -- VERDICT r5 item 4a.  The only a-18 output bound so far was measured on a 0..255 noise image (gradient up to 255 grey levels / px);
the operator is applied to photographs.  Case: U = demo/demo1/input1.jpg (512 x 512, the decoded array already committed in e2e_demo_512.npz),
source = the (12+1) x (12+1) = 169-point rigid mesh of core/UDIS2/Homography/network.py:9-10,83 in [-1, 1], target = source + a smooth seeded
perturbation of at most 0.012 (3 px), out_size 512 x 512.
  photo_source / photo_target   the control points
  photo_out_sub                 the reference's output, every 4th pixel in y and x (fp32), + photo_out_sum (fp64 checksum of the full output)
  photo_T                       the solved coefficients (restated with the reference's calls, checked to reproduce the output through the oracle)
  floor_avx2_photo_out_max / _p99 / floor_sse4_2_*   the reference against ITSELF on this input under MKL_ENABLE_INSTRUCTIONS=AVX2 / SSE4_2
"""
from __future__ import annotations

import os
import subprocess
import sys

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")


def case():
    img = np.load(os.path.join(OUT, "e2e_demo_512.npz"))["demo1_input1"]                      # uint8 HWC, the reference's own demo image
    U = torch.from_numpy(img).permute(2, 0, 1).float()[None].contiguous()                     # [1, 3, 512, 512], 0..255 (out.py:137-143)
    gh = gw = 12
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, gh + 1), torch.linspace(-1, 1, gw + 1), indexing="ij")
    src = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1)[None].float().contiguous()        # [1, 169, 2] (x, y)
    g = torch.Generator().manual_seed(606)
    amp = torch.rand(2, 3, generator=g) * 0.012
    ph = torch.rand(2, 3, generator=g) * 6.28
    x, y = src[0, :, 0], src[0, :, 1]
    dx = amp[0, 0] * torch.sin(2.1 * x + ph[0, 0]) * torch.cos(1.3 * y + ph[0, 1]) + 0.3 * amp[0, 2] * torch.sin(3.7 * y + ph[0, 2])
    dy = amp[1, 0] * torch.cos(1.7 * x + ph[1, 0]) * torch.sin(2.3 * y + ph[1, 1]) + 0.3 * amp[1, 2] * torch.cos(2.9 * x + ph[1, 2])
    tgt = (src + torch.stack([dx, dy], -1)[None]).float().contiguous()
    return U, src, tgt


def run_reference():
    from oracle.ref_harness import make_tps_goldens as mk
    import contextlib
    import importlib
    import io
    mk.install_inference_stubs()
    U, src, tgt = case()
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        tps = importlib.import_module("core.udis_utils.torch_tps_transform")
        out = tps.transformer(U, src, tgt, (512, 512))
        B, N, _ = src.shape                                                                    # the reference's T, its own calls (:149-185)
        p = torch.cat([torch.ones(B, N, 1).float(), src], 2)
        d2 = torch.sum(torch.square(p.reshape(B, -1, 1, 3) - p.reshape(B, 1, -1, 3)), 3)
        r = d2 * torch.log(d2 + 1e-6)
        W = torch.cat((torch.cat((p, r), 2), torch.cat((torch.zeros(B, 3, 3).float(), p.permute(0, 2, 1)), 2)), 1)
        Tm = torch.matmul(torch.inverse(W.type(torch.float64)), torch.cat((tgt, torch.zeros(B, 3, 2)), 1).type(torch.float64)).permute(0, 2, 1).float()
    return U, src, tgt, out, Tm


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        _, _, _, out, _ = run_reference()
        np.save(sys.argv[2], out.numpy())
        return
    U, src, tgt, out, Tm = run_reference()
    rec = dict(photo_source=src.numpy(), photo_target=tgt.numpy(), photo_out_sub=out[..., ::4, ::4].numpy().copy(), photo_out_sum=np.array(out.double().sum().item()),
               photo_T=Tm.numpy(), photo_out_absmax=np.array(out.abs().max().item()))
    # the oracle (CPU restatement) must reproduce the reference here too
    from oracle import geom
    o_out, o_T = geom.tps_transformer(U, src, tgt, (512, 512))
    rec["oracle_vs_reference_out_max"] = np.array((o_out - out).abs().max().item())
    rec["oracle_vs_reference_T_rel"] = np.array(((o_T - Tm).abs().max() / Tm.abs().max()).item())
    print("oracle vs reference on the photograph: out max", rec["oracle_vs_reference_out_max"], "T rel", rec["oracle_vs_reference_T_rel"])
    for isa in ("AVX2", "SSE4_2"):
        path = f"/tmp/tps_photo_{isa}.npy"
        env = dict(os.environ, MKL_ENABLE_INSTRUCTIONS=isa)
        subprocess.check_call([sys.executable, "-m", "oracle.ref_harness.make_r6_goldens", "--child", path], env=env)
        d = np.abs(np.load(path) - out.numpy())
        rec[f"floor_{isa.lower()}_photo_out_max"] = np.array(d.max())
        rec[f"floor_{isa.lower()}_photo_out_p99"] = np.array(np.percentile(d, 99))
        print(f"reference vs itself under MKL {isa}: max {d.max():.4e} p99 {np.percentile(d, 99):.4e} grey levels")
    np.savez_compressed(os.path.join(OUT, "tps_photo_512.npz"), **rec)
    print("wrote tests/golden/tps_photo_512.npz", {k: (v.shape if v.ndim else float(v)) for k, v in rec.items()})


if __name__ == "__main__":
    main()
