"""Where does the evaluation harness spend its time?  (decode starvation vs host enqueue vs GPU)"""
import os, sys, tempfile, time
import torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import stitch_amd
from stitch_amd import evaluate as ev
import bench

torch.cuda.set_device(0)
cfg, _ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
torch.manual_seed(1234)
model = stitch_amd.build_model(cfg).cuda().eval()
root = tempfile.mkdtemp()
N = 240
bench.write_jpeg_split(root, N)
ds = ev.UDISDataset(root + "/", phase="testing")
dev = torch.device("cuda", 0)
print("cpus", len(os.sched_getaffinity(0)))
t0 = time.perf_counter(); [ds.load_u8(i) for i in range(20)]; print("decode ms/pair (1 thread)", (time.perf_counter() - t0) / 20 * 1e3)

class Cached:
    def __init__(self, ds): self.items = [ds.load_u8(i) for i in range(len(ds))]
    def __len__(self): return len(self.items)
    def load_u8(self, i): return self.items[i]

def run(d, **kw):
    ev.validate_with_model(model, d, device=dev, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev.validate_with_model(model, d, device=dev, **kw)
    torch.cuda.synchronize()
    return N / (time.perf_counter() - t0)

cached = Cached(ds)
for s in (3, 4):
    print("cached in RAM, streams", s, run(cached, streams=s))
for w in (2, 4, 8, 12):
    model._eval_pipeline = None
    print("jpeg, workers", w, run(ds, streams=3, decode_workers=w))
