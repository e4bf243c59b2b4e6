"""End-to-end throughput of the out.py inference loop (reference out.py:158-312, 351-357): JPEG pairs on disk -> the 10 result JPEGs per pair.
`run_pairs` (pair i + 1's network graph in flight, decode / encode on worker threads) against calling `inference_one_data` pair by pair.

    python tools/bench_out_harness.py [N=48] > profiles/r4_out_harness.json
"""
import importlib.util, json, os, sys, tempfile, time
import numpy as np
import torch
ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import stitch_amd
from stitch_amd.data import structured_pair
from PIL import Image

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
spec_ = importlib.util.spec_from_file_location("stitch_out_harness_b", os.path.join(ROOT, "out.py"))
outmod = importlib.util.module_from_spec(spec_)
spec_.loader.exec_module(outmod)
root = tempfile.mkdtemp()
names = []
base = [structured_pair(512, 512, seed=40 + i) for i in range(8)]
for i in range(N):
    d = os.path.join(root, "demo", f"p{i:03d}")
    os.makedirs(d)
    a, b = base[i % 8]
    for n, t in (("input1.jpg", a), ("input2.jpg", b)):
        arr = np.roll(t[0].permute(1, 2, 0).numpy().astype(np.uint8), (3 * (i // 8), -5 * (i // 8)), (0, 1))
        Image.fromarray(arr).save(os.path.join(d, n), quality=95)
    names.append(f"p{i:03d}/")
open(os.path.join(root, "demo", "demo.txt"), "w").write("\n".join(names) + "\n")
cfg = outmod.get_config(["--data_root_path", os.path.join(root, "demo") + "/"])
todo = outmod.get_data_dict_list(cfg.data_root_path, cfg.txt_file)
torch.manual_seed(1234)
model = stitch_amd.build_model(cfg).cuda().eval()
comp = stitch_amd.composition.Network().cuda().eval()
inp = outmod.load_inpainter("passthrough_inpainter")
import contextlib, io
res = {}
with contextlib.redirect_stdout(io.StringIO()):
    for tag, fn in (("pipelined_depth3", lambda d_: outmod.run_pairs(cfg, todo, d_, model, comp, inp, depth=3)),
                    ("pipelined", lambda d_: outmod.run_pairs(cfg, todo, d_, model, comp, inp)),
                    ("pipelined_depth4", lambda d_: outmod.run_pairs(cfg, todo, d_, model, comp, inp, depth=4)),
                    ("pair_by_pair", lambda d_: [outmod.inference_one_data(cfg, dd, d_, model, comp, inp) for dd in todo])):
        for rep in range(2):                    # first pass: graph capture, file cache, lazy constants
            dst = os.path.join(root, f"{tag}{rep}") + "/"
            os.makedirs(dst)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(dst)
            torch.cuda.synchronize()
            res[tag] = N / (time.perf_counter() - t0)
print(json.dumps({"pairs": N, "what": "out.py loop end to end on synthetic 512x512 JPEG pairs: decode, forward(test_out), TPS post-pipeline (opencv-like back-end, "
                  "mix_fn all_img1_with_inpaint with the pass-through inpainter), composition network, 10 JPEG files per pair written",
                  "pairs_per_s_pipelined_run_pairs": res["pipelined"], "depth": 2, "pairs_per_s_depth3": res["pipelined_depth3"], "pairs_per_s_depth4": res["pipelined_depth4"], "pairs_per_s_pair_by_pair": res["pair_by_pair"], "weights": "random init"}))
