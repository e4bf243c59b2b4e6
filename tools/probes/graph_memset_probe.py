"""Minimal reproducer of round 4's finding (csrc/geom.hip, st_range_map): is a `hipMemsetAsync` captured into a hipGraph executed on REPLAYS?

    python tools/probes/graph_memset_probe.py        (GPU box; prints the ROCm / HIP runtime versions and the verdict)

Capture `hipMemsetAsync(buf, 0, n)` followed by `buf += 1` (a kernel), replay three times: a working memset node leaves 1.0 after every replay, a
skipped one 2.0, 3.0, ...  Variants: 32-bit and 64-bit element buffers (the range map's accumulator is u64), a buffer allocated before the capture and one
allocated inside it (graph-private pool), `hipMemsetD32Async`.  ADVICE r4 asked for the cause to be pinned down; this is the record."""
import ctypes as C
import sys
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemsetD32Async.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
ver = C.c_int(0)
hip.hipRuntimeGetVersion(C.byref(ver))
print(f"torch {torch.__version__}, HIP runtime {ver.value}, device {torch.cuda.get_device_name(0)}")


def run(tag, dtype, n, inside, d32=False):
    buf = None if inside else torch.full((n,), 5, dtype=dtype, device="cuda")
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        if inside:
            buf = torch.empty((n,), dtype=dtype, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        if d32:
            rc = hip.hipMemsetD32Async(buf.data_ptr(), 0, buf.numel() * buf.element_size() // 4, st)
        else:
            rc = hip.hipMemsetAsync(buf.data_ptr(), 0, buf.numel() * buf.element_size(), st)
        assert rc == 0, rc
        buf.add_(1)
    seen = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        seen.append((float(buf.min()), float(buf.max())))
    ok = all(v == (1.0, 1.0) for v in seen)
    print(f"{tag:58s} after replays 1..3 (min, max): {seen}  -> {'memset node runs on every replay' if ok else 'MEMSET NODE NOT EXECUTED ON REPLAYS'}")
    return ok


res = [run("hipMemsetAsync, fp32 [4096], buffer from before the capture", torch.float32, 4096, False),
       run("hipMemsetAsync, int64 [512*512], buffer from before", torch.int64, 512 * 512, False),
       run("hipMemsetAsync, int64 [512*512], buffer allocated in capture", torch.int64, 512 * 512, True),
       run("hipMemsetD32Async, int64 [512*512], buffer from before", torch.int64, 512 * 512, False, d32=True),
       run("hipMemsetAsync, int64 [1024*1024+3], odd size", torch.int64, 1024 * 1024 + 3, False)]
print("all variants clear on every replay" if all(res) else "at least one variant is not cleared on replays: keep accumulators zeroed by kernels (csrc/geom.hip zero_u64_kernel)")
sys.exit(0)
