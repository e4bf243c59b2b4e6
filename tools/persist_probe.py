"""Do two GEMM kernels from different streams share the CUs better when each holds ONE workgroup slot per CU?
M = 8192 GEMMs (the decoder's shape class) launch 512 workgroups = one round at two per CU; with ST_PERSIST_SLOTS=256 the plain-matrix
path walks two M tiles per workgroup on a 256-workgroup grid (one per CU), leaving the second slot to another stream's kernel.
    ST_PERSIST_SLOTS=512 python tools/persist_probe.py ; ST_PERSIST_SLOTS=256 python tools/persist_probe.py"""
import os, sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import stitch_amd
ops = stitch_amd.ops
shapes = [(8192, 256, 1920), (8192, 256, 1152), (8192, 128, 2304), (8192, 256, 256), (8192, 384, 1280)]
for nstreams in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    bufs = [[(torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / K ** 0.5, torch.empty(M, N, device="cuda")) for (M, N, K) in shapes]
            for _ in range(nstreams)]
    def run(reps):
        for s, bs in zip(streams, bufs):
            with torch.cuda.stream(s):
                for _ in range(reps):
                    for a, w, c in bs:
                        ops.conv_gemm(a, w, c)
    run(2); torch.cuda.synchronize()
    graphs = []
    for s, bs in zip(streams, bufs):             # one graph per stream: no host launch cost in the measurement
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(8):
                for a, w, c in bs:
                    ops.conv_gemm(a, w, c)
        graphs.append(g)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    def replay(n):
        for _ in range(n):
            for s, g in zip(streams, graphs):
                with torch.cuda.stream(s):
                    g.replay()
    replay(2); torch.cuda.synchronize()
    t0.record()
    for s in streams: s.wait_event(t0)
    replay(10)
    for s in streams: torch.cuda.current_stream().wait_stream(s)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1)
    fl = sum(2 * M * N * K for M, N, K in shapes) * 8 * 10 * nstreams
    print(f"ST_PERSIST_SLOTS={os.environ.get('ST_PERSIST_SLOTS', '512')} streams {nstreams}: {fl / ms / 1e9:.1f} TFLOP/s ({ms / (8 * 10 * nstreams * len(shapes)) * 1e3:.1f} us per GEMM)")
