"""Diagnostic: which output of a captured forward changes after an eager call / after NaN-filling freed memory."""
import os, sys, tempfile
import torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import stitch_amd
from stitch_amd import evaluate as ev, ops
from stitch_amd.data import structured_pair

mode = sys.argv[1]
torch.cuda.set_device(0)
cfg, _ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
torch.manual_seed(1234)
model = stitch_amd.build_model(cfg).cuda().eval()
dev = torch.device("cuda", 0)
a, b = [t.cuda() for t in structured_pair(512, 512, seed=40)]
a2, b2 = [t.cuda() for t in structured_pair(512, 512, seed=41)]
g = model.graphed("test_eval")
o = g(a, b)
torch.cuda.synchronize()
ref = {k: (v[0] if isinstance(v, list) else v).clone() for k, v in o.items()}
def cmp(tag):
    o = g(a, b)
    torch.cuda.synchronize()
    print(tag, {k: f"{((v[0] if isinstance(v, list) else v) - ref[k]).abs().max().item():.2e}" for k, v in o.items()})
cmp("replay")
if mode == "nanfill":
    xs = [torch.full((256 * 1024 * 1024,), float("nan"), device=dev) for _ in range(8)]     # 8 GiB of fresh + cached blocks
    del xs
elif mode == "homo":
    model.predict_homo(a2, b2)
elif mode == "prep":
    x = torch.empty((2 * 512 * 512, 4), device=dev)
    ops.prep_image(a2.contiguous(), x[:512 * 512], 4, 1.0, 127.5, 1.0)
elif mode == "eager_same":
    model(a, b, type="test_eval")
torch.cuda.synchronize()
cmp("after " + mode)
oe = model(a, b, type="test_eval")
print("eager vs first replay", {k: f"{((v[0] if isinstance(v, list) else v) - ref[k]).abs().max().item():.2e}" for k, v in oe.items()})
