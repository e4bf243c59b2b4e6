"""Upper bounds by ablation: what the bench would read if a kernel cost NOTHING (its launch is skipped; results are then wrong, only
the timing is of interest).  python tools/ablate.py name[,name...] [bench args]   -- names are functions of stitch_amd.ops that return None
(decoder_token_chain, cost_lookup9x9, flow_encode, layernorm, window_attention, attention_kvlds, attention_small, latent_pool, sine_pe,
dwconv3x3_residual, gma_aggregate, sepconv_gru, ...) or 'narrow' (conv_gemm calls with N <= 4)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stitch_amd  # noqa: E402
import bench  # noqa: E402

names = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] != "none" else []
ops = stitch_amd.ops
for n in names:
    if n == "narrow":
        real = ops.conv_gemm
        def conv_gemm(x, w, out, **kw):
            if out.shape[1] <= 4:
                return None
            return real(x, w, out, **kw)
        ops.conv_gemm = conv_gemm
    else:
        assert hasattr(ops, n), n
        setattr(ops, n, (lambda *a, **k: a[1]) if n == "linear_chain128" else (lambda *a, **k: None))
sys.argv = [sys.argv[0]] + (sys.argv[2:] or ["--steps", "80", "--warmup", "10", "--no-cpu-baseline", "--no-corr-roofline"])
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
for l in buf.getvalue().splitlines():
    if l.startswith("{"):
        d = json.loads(l)
        print(f"ablate {','.join(names) or 'none':40s} pairs/s {d['value']:.2f}  1-in-flight {d.get('value_1_in_flight') or 0:.2f}")
