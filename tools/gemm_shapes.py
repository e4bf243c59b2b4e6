"""Per-shape time breakdown of the st_conv_gemm launches of one test_eval step (HIP events, library observer)."""
import sys, collections, ctypes as C, torch
sys.path.insert(0, '/root/repo')
import stitch_amd
from stitch_amd import data as inputs
lib, GemmDesc = stitch_amd._lib.lib, stitch_amd._lib.GemmDesc
cfg, _ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
torch.manual_seed(1234)
model = stitch_amd.build_model(cfg).cuda().eval()
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pairs = [inputs.structured_pair(512, 512, seed=7 + i) for i in range(NB)]
a, b = torch.cat([p[0] for p in pairs]).cuda(), torch.cat([p[1] for p in pairs]).cuda()
for _ in range(2): model(a, b, type="test_eval")
rec, open_ev = [], []

@C.CFUNCTYPE(None, C.POINTER(GemmDesc), C.c_void_p, C.c_int32, C.c_void_p)
def observer(desc, stream, phase, user):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream())
    if phase == 0:
        d = desc.contents
        bt = max(1, d.batch)
        open_ev.append(((d.M, d.N, d.K, f"{d.kh}x{d.kw}s{d.sh}", bt, d.epi), 2.0 * d.M * d.N * d.K * bt, ev))
    else:
        key, fl, e0 = open_ev.pop()
        rec.append((key, fl, e0, ev))

lib.st_set_gemm_observer(C.cast(observer, C.c_void_p), None)
model(a, b, type="test_eval"); torch.cuda.synchronize()
lib.st_set_gemm_observer(None, None)
agg = collections.OrderedDict()
for key, fl, e0, e1 in rec:
    t = e0.elapsed_time(e1)
    c = agg.setdefault(key, [0, 0.0, 0.0]); c[0] += 1; c[1] += t; c[2] += fl
tot = sum(v[1] for v in agg.values())
print(f"total gemm ms {tot:.2f} over {len(rec)} launches, {sum(v[2] for v in agg.values())/1e9:.1f} GFLOP")
for key, (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:48]:
    print(f"M={key[0]:>8} N={key[1]:>5} K={key[2]:>5} {key[3]:>6} b={key[4]} epi={key[5]} calls={n:>4} ms={t:7.3f} ({100*t/tot:4.1f}%) avg_us={1e3*t/n:8.1f} TF={fl/t/1e9:6.1f}")
