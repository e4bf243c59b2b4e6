"""Per-stage parity trace of the UDIS2 homography regression (network.py:121-199): HIP vs the CPU oracle in fp32 and fp64
(test tooling; same three-way reading as tools/trace_parity.py).   python tools/trace_parity_homo.py [seed]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from oracle import geom, nets, spec  # noqa: E402
import stitch_amd  # noqa: E402
from stitch_amd.data import structured_pair  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item(), (a - b).abs().max().item()


def oracle_stages(w, a, b):
    x1, x2 = a / 127.5 - 1.0, b / 127.5 - 1.0
    f1 = nets.resnet_stage2(w, nets.resnet_stage1(w, x1))
    f2 = nets.resnet_stage2(w, nets.resnet_stage1(w, x2))
    c = nets.ccl(f1, f2)
    return f1, f2, c, nets.regress(w, c)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    sd = spec.seeded_state_dict(1234)
    a, b = structured_pair(512, 512, seed=seed)
    cfg, _ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
    model = stitch_amd.build_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    hb, ops = model.homo_backbone, stitch_amd.ops
    hb.pack()
    B, H, W = 1, 512, 512
    with torch.no_grad():
        x = torch.empty((2 * H * W, 4), device="cuda")
        ops.prep_image(a.cuda(), x[:H * W], 4, 1.0, 127.5, 1.0)
        ops.prep_image(b.cuda(), x[H * W:], 4, 1.0, 127.5, 1.0)
        f, h, w = hb.features(x, 2, H, W)
        c = hb.ccl(f, 1, h, w)
        off = hb.regress(c, 1, h, w)
        o32 = oracle_stages(nets.W(sd, "homo_backbone."), a, b)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        torch.set_default_dtype(torch.float64)
        o64 = oracle_stages(nets.W(sd64, "homo_backbone."), a.double(), b.double())
        torch.set_default_dtype(torch.float32)
    fh = f.view(2, h, w, -1).permute(0, 3, 1, 2)
    ch = c[:, :2].reshape(1, h, w, 2).permute(0, 3, 1, 2)
    print(f"# UDIS2 homography net 512x512 (seed {seed})")
    print(f"# {'stage':28s} {'hip-o32 rel':>12s} {'hip-o64 rel':>12s} {'o32-o64 rel':>12s}   {'hip-o32 abs':>12s} {'hip-o64 abs':>12s} {'o32-o64 abs':>12s}")
    for name, hv, a32, a64 in (("resnet stage2 (img1)", fh[0:1], o32[0], o64[0]), ("resnet stage2 (img2)", fh[1:2], o32[1], o64[1]),
                               ("CCL soft-argmax flow", ch, o32[2], o64[2]), ("corner offsets (px)", off, o32[3], o64[3])):
        r1, a1 = rel(hv, a32)
        r2, a2 = rel(hv, a64)
        r3, a3 = rel(a32, a64)
        print(f"  {name:28s} {r1:12.3e} {r2:12.3e} {r3:12.3e}   {a1:12.3e} {a2:12.3e} {a3:12.3e}")
    src = torch.tensor([[0., 0.], [512., 0.], [0., 512.], [512., 512.]])[None]
    Hh = geom.dlt4(src / 8, (src + off.cpu().reshape(1, 4, 2)) / 8)
    H32 = geom.dlt4(src / 8, (src + o32[3].reshape(1, 4, 2)) / 8)
    H64 = geom.dlt4_torch((src / 8).double(), ((src + o64[3].reshape(1, 4, 2)) / 8).double())
    print(f"# H (DLT of those offsets): hip-o32 {(Hh - H32).abs().max():.3e}  hip-o64 {(Hh.double() - H64).abs().max():.3e}  o32-o64 {(H32.double() - H64).abs().max():.3e}")


if __name__ == "__main__":
    main()
