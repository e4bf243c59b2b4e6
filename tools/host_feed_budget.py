"""Host side of the evaluation harness at 8 ranks: what one pair costs in JPEG decode + staging + H2D per host core, how many
cores 8 ranks x the measured per-GPU rate need, what this box offers.  No model, no forward: only the feed.
    python tools/host_feed_budget.py [pairs_per_s_per_gpu=86]  > profiles/r6_host_feed_budget.txt"""
import os, sys, tempfile, time, shutil
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from stitch_amd import evaluate as sev

rate = float(sys.argv[1]) if len(sys.argv) > 1 else 86.0
root = os.path.join(tempfile.gettempdir(), "stitch_feed_budget")
shutil.rmtree(root, ignore_errors=True)
N = 192
bench.write_jpeg_split(root, N)
ds = sev.UDISDataset(root + "/", phase="testing")
try:
    aff = len(os.sched_getaffinity(0))
except AttributeError:
    aff = os.cpu_count()
print(f"# host feed budget of stitch_amd.evaluate (JPEG 512x512 quality 95 pairs on local disk, page cache warm), {N} pairs")
print(f"os.cpu_count() = {os.cpu_count()}, cores this process may use (sched_getaffinity) = {aff}")
for i in range(N):
    ds.load_u8(i)                                   # page the files in
t0 = time.perf_counter()
for i in range(N):
    ds.load_u8(i)
dec = (time.perf_counter() - t0) / N
print(f"decode (PIL open + RGB + ndarray, both images of a pair), ONE thread: {dec * 1e3:.2f} ms per pair = {1 / dec:.0f} pairs/s per core")
for th in (2, 4, 8, 16):
    with ThreadPoolExecutor(th) as ex:
        t0 = time.perf_counter()
        list(ex.map(ds.load_u8, range(N)))
        dt = (time.perf_counter() - t0) / N
    print(f"  {th:2d} decode threads: {1 / dt:7.0f} pairs/s ({dt * 1e3 * th:.2f} thread-ms per pair: PIL releases the GIL while it decodes)")
a, b = ds.load_u8(0)
pin = [torch.empty((1, 512, 512, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
pn = [p.numpy() for p in pin]
t0 = time.perf_counter()
for _ in range(2000):
    np.copyto(pn[0][0], a); np.copyto(pn[1][0], b)
cp = (time.perf_counter() - t0) / 2000
print(f"staging copy into the slot's pinned buffers (2 x 0.75 MB): {cp * 1e3:.3f} ms per pair")
if torch.cuda.is_available():
    devb = [torch.empty((1, 512, 512, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500):
        devb[0].copy_(pin[0], non_blocking=True); devb[1].copy_(pin[1], non_blocking=True)
    e1.record(); torch.cuda.synchronize()
    h2d = e0.elapsed_time(e1) / 500
    print(f"pinned H2D of a pair (1.5 MB) on the copy stream: {h2d * 1e3:.1f} us = {1.5 / h2d:.1f} GB/s per rank; 8 ranks x {rate:.0f} pairs/s = {8 * rate * 1.5 / 1e3:.2f} GB/s of host reads in total")
need = 8 * rate * (dec + cp)
print(f"8 ranks x {rate:.0f} pairs/s = {8 * rate:.0f} pairs/s need {need:.1f} host cores of decode + staging (+ one launching thread per rank = {need + 8:.1f});")
print(f"this box grants {aff} cores to a 1-GPU job: one rank's feed ({rate:.0f} pairs/s) takes {rate * (dec + cp):.2f} cores -- the harness's 4 decode threads per rank cover it "
      f"with {4 / (rate * (dec + cp)):.1f}x headroom; an 8-GPU node must offer >= {int(need + 8) + 1} cores to the job or the feed, not the GPUs, sets the rate.")
shutil.rmtree(root, ignore_errors=True)
