"""evaluate.py's batch size (12 pairs per forward): plumbing check of the whole path at B=12 (finite outputs, time)."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import stitch_amd
from stitch_amd import data as inputs
cfg, _ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
torch.manual_seed(1234)
model = stitch_amd.build_model(cfg).cuda().eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
pairs = [inputs.structured_pair(512, 512, seed=60 + i, shift=(2 * (i % 5) - 3, 5 - (i % 7))) for i in range(B)]
A, Bm = torch.cat([p[0] for p in pairs]).cuda(), torch.cat([p[1] for p in pairs]).cuda()
o = model(A, Bm, type="test_eval"); torch.cuda.synchronize()
t0 = time.time(); o = model(A, Bm, type="test_eval"); torch.cuda.synchronize(); dt = time.time() - t0
print(f"B={B}: {dt*1e3:.1f} ms per forward = {B/dt:.1f} pairs/s; finite {bool(torch.isfinite(o['final_warp_output']).all())}; "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
s = model(pairs[5][0].cuda(), pairs[5][1].cuda(), type="test_eval")
print("sample 5 vs alone: H", (o["H"][5] - s["H"][0]).abs().max().item(), "output_H", (o["output_H"][5] - s["output_H"][0]).abs().max().item())
