"""Per-shape time breakdown of the st_conv_gemm launches of one test_eval step (HIP events)."""
import sys, collections, torch
sys.path.insert(0, '/root/repo')
import stitch_amd
from oracle import inputs
ops = stitch_amd.ops
cfg, _ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
torch.manual_seed(1234)
model = stitch_amd.build_model(cfg).cuda().eval()
a, b = inputs.structured_pair(512, 512, seed=7); a, b = a.cuda(), b.cuda()
for _ in range(2): model(a, b, type="test_eval")
rec = []
orig = ops.conv_gemm
def timed(x, w, out, **kw):
    geom = kw.get("geom"); Cin = x.shape[1]
    if geom is None:
        M = kw.get("M") or x.shape[0]; K = Cin; g = "1x1"
    else:
        B, H, W, kh, kw_, sh, sw, ph, pw = geom[:9]
        Ho, Wo = ((H + 2*ph - kh)//sh + 1, (W + 2*pw - kw_)//sw + 1) if len(geom) == 9 else geom[9:11]
        M, K = B*Ho*Wo, kh*kw_*Cin; g = f"{kh}x{kw_}s{sh}"
    batch = max(1, kw.get("batch", 1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(x, w, out, **kw); e1.record()
    rec.append(((M, w.shape[0], K, g, batch), 2.0*M*w.shape[0]*K*batch, e0, e1)); return r
ops.conv_gemm = timed
model(a, b, type="test_eval"); torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, fl, e0, e1 in rec:
    t = e0.elapsed_time(e1)
    c = agg.setdefault(key, [0, 0.0, 0.0]); c[0] += 1; c[1] += t; c[2] += fl
tot = sum(v[1] for v in agg.values())
print(f"total gemm ms {tot:.2f} over {len(rec)} launches")
for key, (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"M={key[0]:>8} N={key[1]:>5} K={key[2]:>5} {key[3]:>6} b={key[4]} calls={n:>4} ms={t:7.3f} ({100*t/tot:4.1f}%) avg_us={1e3*t/n:8.1f} TF={fl/t/1e9:6.1f}")
