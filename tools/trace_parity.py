"""Per-stage parity trace of FlowFormer++: HIP path vs the CPU oracle in fp32 AND in fp64 (test tooling).

    python tools/trace_parity.py [H W seed] > profiles/r2_parity_trace_<H>x<W>.txt

For every stage it prints three distances (max-abs relative to the stage's max magnitude):
    hip-o32   HIP kernels against the fp32 oracle (the reference's arithmetic, pinned bit-for-bit)
    hip-o64   HIP kernels against the same algorithm evaluated in fp64 ("truth")
    o32-o64   the fp32 oracle's own rounding error against that truth
hip-o64 <= ~o32-o64 at a stage means the HIP result is as close to the exact answer as the reference is:
what separates HIP from the reference there is summation-order noise, not a defect.
Imports oracle/ (checker); not part of the product path.
"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from oracle import nets, spec  # noqa: E402
import stitch_amd  # noqa: E402
from stitch_amd.data import structured_pair  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item(), (a - b).abs().max().item()


def main():
    H, Wd, seed = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (96, 128, 3)
    shift = (2, -3) if H < 256 else (5, -9)
    sd = spec.seeded_state_dict(1234)
    a, b = structured_pair(H, Wd, seed=seed, shift=shift)
    cfg, _ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
    model = stitch_amd.build_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    fb = model.flow_backbone
    tr_h = []
    with torch.no_grad():
        up_h, _, (B, H1, W1) = fb.flow_rows(a.cuda(), b.cuda(), trace=tr_h)
    torch.cuda.synchronize()
    N = H1 * W1
    t0 = time.time()
    tr32 = []
    with torch.no_grad():
        up32, _ = nets.flowformer(nets.W(sd, "flow_backbone."), a, b, trace=tr32)
    t32 = time.time() - t0
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    torch.set_default_dtype(torch.float64)
    t0 = time.time()
    tr64 = []
    with torch.no_grad():
        up64, _ = nets.flowformer(nets.W(sd64, "flow_backbone."), a.double(), b.double(), trace=tr64)
    torch.set_default_dtype(torch.float32)
    print(f"# FlowFormer++ {H}x{Wd} (seed {seed}); oracle fp32 {t32:.1f} s, fp64 {time.time() - t0:.1f} s")
    print(f"# {'stage':28s} {'hip-o32 rel':>12s} {'hip-o64 rel':>12s} {'o32-o64 rel':>12s}   {'hip-o32 abs':>12s} {'hip-o64 abs':>12s} {'o32-o64 abs':>12s}")

    def row(name, h, o32, o64):
        r1, a1 = rel(h, o32)
        r2, a2 = rel(h, o64)
        r3, a3 = rel(o32, o64)
        print(f"  {name:28s} {r1:12.3e} {r2:12.3e} {r3:12.3e}   {a1:12.3e} {a2:12.3e} {a3:12.3e}")

    def nchw(rows, C):
        return rows.reshape(B, H1, W1, C).permute(0, 3, 1, 2)

    e_h, e32, e64 = tr_h[0], tr32[0], tr64[0]
    row("context (cnet twins)", nchw(e_h["context"], 256), e32["context"], e64["context"])
    row("fnet(image1)", nchw(e_h["feats"][0], 256), e32["feat_s"], e64["feat_s"])
    row("fnet(image2)", nchw(e_h["feats"][1], 256), e32["feat_t"], e64["feat_t"])
    cm32 = nets.corr_volume(e32["feat_s"], e32["feat_t"]).reshape(B * N, N)
    cm64 = nets.corr_volume(e64["feat_s"], e64["feat_t"]).reshape(B * N, N)
    row("cost volume", e_h["cost_maps"], cm32, cm64)
    mem_h = (e_h["mem"] if e_h["short"] is None else e_h["mem"] + e_h["short"]).reshape(B * N, 8, 128)      # (the short-cut is added in the last layer's epilogue now)
    row("cost memory (encoder out)", mem_h, e32["cost_memory"], e64["cost_memory"])
    c0 = nets.coords_grid(B, H1, W1)
    acc32 = torch.zeros(B, 2, H1, W1)
    acc64 = torch.zeros(B, 2, H1, W1, dtype=torch.float64)
    for it in range(1, len(tr32)):
        th, t32_, t64_ = tr_h[it], tr32[it], tr64[it]
        acc32 = acc32 + t32_["dflow"]
        acc64 = acc64 + t64_["dflow"]
        fl_h = nchw(th["coords1"], 2).cpu() - c0
        row(f"iter {it:2d} cost_forward (81)", nchw(th["corr"][:, :81], 81), t32_["cost_forward"], t64_["cost_forward"])
        row(f"iter {it:2d} cost_global", nchw(th["corr"][:, 84:148], 64), t32_["cost_global"], t64_["cost_global"])
        row(f"iter {it:2d} net", nchw(th["net"], 128), t32_["net"], t64_["net"])
        row(f"iter {it:2d} flow (1/8 res, px/8)", fl_h, acc32, acc64)
    row("flow_up (full res, px)", up_h, up32, up64)
    d = (up_h.cpu().double() - up64).abs().flatten()
    d2 = (up32.double() - up64).abs().flatten()
    d3 = (up_h.cpu() - up32).abs().flatten()
    k = int(0.99 * d.numel())
    print(f"# flow_up px: hip-o32 max {d3.max():.3e} p99 {d3.kthvalue(k).values:.3e} | hip-o64 max {d.max():.3e} p99 {d.kthvalue(k).values:.3e} "
          f"| o32-o64 max {d2.max():.3e} p99 {d2.kthvalue(k).values:.3e}")


if __name__ == "__main__":
    main()
