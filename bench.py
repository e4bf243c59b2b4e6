"""Headline benchmark: stitched image-pairs/s at 512x512 (BASELINE.json metric) on N MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload 512|1024]

N > 1: if the process was not started by ``torch.distributed.run`` it starts
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`` itself (as a child: this parent
never touches the GPU) and relays rank 0's JSON line; under the driver's own torchrun launch the ranks run directly.

A step = one pass of the hot path over one synthetic pair:
  --workload 512  (default, BASELINE.json configs[1]): ``FlowHomoAdpater.forward(type="test_eval")`` on a 512x512 pair,
                  batch 1 (1 homography pass, 2 FlowFormer++ passes, warp / occlusion / blend), hipGraph replay, 3 pairs in
                  flight on 3 HIP streams; the single-stream figure is reported beside it (``value_1_in_flight``)
  --workload 1024 (configs[3]): ``forward(type="test_out")`` on 1024x1024 pairs, 4 distinct pairs per GPU in turn, batch 1
                  (the canvas is per pair and read back to the host mid-way, so this path is eager, one stream)
Pairs are independent: ranks shard them with no data-path collective (weak scaling: every rank runs K steps); one RCCL
all-gather moves the per-pair PSNR at the end.  Inputs and random-init weights are resident in HBM before the timed region.

Besides the contract fields the JSON line carries
  roofline      -- the dominant kernel family (fp32-MFMA implicit GEMM): algorithmic FLOPs of its launches in one step /
                   their summed HIP-event durations vs the 157.3 TFLOP/s fp32 matrix peak; measured on an instrumented step
                   after the timed region (events on the stream the kernels are launched on); ``traffic`` = HBM bytes per
                   STEP of that family from the committed PMC passes, ``algorithmic_bytes`` = sum of A+W+C of its launches
  corr_volume   -- the all-pairs correlation kernel alone (B=8, configs[2]) against both roofs
  cpu_baseline  -- the CPU oracle (torch-CPU port of the reference path) timed on this host, rank 0, N=1
  parity        -- HIP output vs that oracle output on the same pair: dPSNR / dSSIM of the evaluate.py metric, flow error
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = vector fp32 peak
HBM_PEAK_GBS = 8000.0
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16: 32 cycles per SIMD), no sparsity


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=("512", "1024"), default="512")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-corr-roofline", action="store_true", help="skip the stand-alone corr-volume timing (PMC passes)")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying the hipGraph")
    ap.add_argument("--streams", type=int, default=3, help="independent forwards in flight per GPU (one hipGraph + HIP stream each)")
    ap.add_argument("--batch", type=int, default=1, help="pairs per forward: 1 = BASELINE configs[1] (default), 8 = configs[2]")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the ranks (nccl = RCCL; gloo for CPU rehearsal of the launcher)")
    ap.add_argument("--dry-run", action="store_true", help="launcher/collective plumbing only: no GPU, no model (tests/test_dist_cpu.py)")
    ap.add_argument("--harness", choices=("none", "eval"), default="eval",
                    help="eval (default, workload 512): also time the PRODUCT harness end to end -- stitch_amd.evaluate.validate_with_model over "
                         "--harness-pairs synthetic 512x512 JPEG pairs written to a temp dir (JPEG decode + H2D + forward + metric + final "
                         "copy inside the clock) -> harness_pairs_per_s beside value")
    ap.add_argument("--harness-pairs", type=int, default=240)
    ap.add_argument("--harness-batch", type=int, default=4,
                    help="the harness is timed twice: one pair per forward (like `value`) and batches of this size (the reference's loader "
                         "batches 12, evaluate.py:34) -> harness_batched_pairs_per_s; 0 or 1 = skip the batched pass")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a 1-GPU box: every rank computes on cuda:0 (use with --backend gloo; RCCL needs one GPU per rank)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------- launcher (no GPU use)
def launcher_command(args, port):
    """The torchrun command line the parent starts for ``--gpus N`` (kept separate so the CPU test can check it)."""
    fwd = ["--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--workload", args.workload,
           "--streams", str(args.streams), "--batch", str(args.batch), "--backend", args.backend, "--harness", args.harness,
           "--harness-pairs", str(args.harness_pairs), "--harness-batch", str(args.harness_batch)]
    for flag, on in (("--no-cpu-baseline", args.no_cpu_baseline), ("--no-corr-roofline", args.no_corr_roofline),
                     ("--eager", args.eager), ("--dry-run", args.dry_run), ("--share-gpu", args.share_gpu)):
        if on:
            fwd.append(flag)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + fwd


def launch_ranks(args):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(launcher_command(args, port), env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if proc.returncode != 0 or line is None:
        sys.stderr.write(proc.stdout)
        raise SystemExit(proc.returncode or 1)
    print(line, flush=True)


# ---------------------------------------------------------------------------------------------- measurements
def instrumented_step(run_step):
    """Run one step with a HIP-event pair around every st_conv_gemm launch (library observer hook, so the GEMMs enqueued by
    the operator-level entry points are seen too); returns (flops, ms, launches, algorithmic bytes)."""
    import ctypes as C
    import torch
    import stitch_amd
    lib, GemmDesc = stitch_amd._lib.lib, stitch_amd._lib.GemmDesc
    rec, open_ev = [], []
    plan = (C.c_int32 * 4)()

    @C.CFUNCTYPE(None, C.POINTER(GemmDesc), C.c_void_p, C.c_int32, C.c_void_p)
    def observer(desc, stream, phase, user):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream())
        if phase == 0:
            open_ev.append(ev)
        else:
            d = desc.contents
            lib.st_gemm_last_plan(plan)
            nb = max(1, d.batch)
            a_rows = d.M if d.kh * d.kw <= 1 else (d.M // max(1, d.Ho * d.Wo)) * d.H * d.W       # conv: input pixels
            if plan[0] in (5, 6):       # fused row kernels, reported as M x 128 L x 128 / M x 2 hidden x 128: rows in, weights, rows out
                abytes = 4.0 * (2 * d.M * 128 + d.N * d.K) + (4.0 * d.M * 128 if plan[0] == 6 else 0.0)      # (+ the MLP's residual re-read)
            elif plan[0] == 9:          # the same block tail on the split3 kernel (st_mlp128_split3): fp32 rows in (+ the projection's residual) and out, weights as three bf16 planes
                abytes = 4.0 * (3 * d.M * 128) + 6.0 * d.N * d.K
            elif d.split3:              # exact-split operands: three bf16 planes = 6 bytes per element of A and W; C fp32 (unless c_no_f32)
                abytes = nb * (6.0 * (a_rows * d.Cin + d.N * d.K) + (0.0 if d.c_no_f32 else 4.0 * d.M * d.N))
            else:
                abytes = 4.0 * nb * (a_rows * d.Cin + d.N * d.K + d.M * d.N * (2 if d.c_t else 1))          # A + W + C (+ transposed copy), fp32
            if d.c_planes:              # the result also leaves as planes
                abytes += 6.0 * nb * d.M * (d.N // 2 if d.epi == 5 else d.N)
            rec.append((2.0 * d.M * d.N * d.K * nb, abytes, open_ev.pop(), ev, bool(d.split3)))

    lib.st_set_gemm_observer(C.cast(observer, C.c_void_p), None)
    st = torch.cuda.current_stream()
    empties = []
    try:
        # hold the stream back (~0.1 s spin kernel) while the host enqueues the whole eager step: the GPU then runs the ~1000
        # launches back to back and a bracket reads its kernel(s) + the event overhead, not the Python launch latency that an idle
        # GPU would otherwise wait for in front of every short kernel
        torch.cuda._sleep(int(2.4e8))
        run_step()
        # what an EMPTY event bracket reads (two back-to-back records are a few us apart on the GPU timeline): every bracket above
        # carries that on top of its kernel(s); subtract the median
        for _ in range(64):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            e1.record(st)
            empties.append((e0, e1))
        torch.cuda.synchronize()
    finally:
        lib.st_set_gemm_observer(None, None)
    ov = sorted(e0.elapsed_time(e1) for e0, e1 in empties)[len(empties) // 2]
    raw = [r[2].elapsed_time(r[3]) for r in rec]
    s3 = [(r, t) for r, t in zip(rec, raw) if r[4]]
    return dict(flops=sum(r[0] for r in rec), ms_raw=sum(raw), ms=sum(max(0.0, t - ov) for t in raw), launches=len(rec),
                alg_bytes=sum(r[1] for r in rec), bracket_overhead_us=1e3 * ov,
                split3_flops=sum(r[0] for r, _ in s3), split3_ms=sum(max(0.0, t - ov) for _, t in s3), split3_launches=len(s3))


def provenance(path):
    """file name + mtime + content hash of a committed profile summary quoted in the line: a stale file cannot be quoted silently."""
    import hashlib
    import datetime
    raw = open(path, "rb").read()
    return {"file": os.path.relpath(path, ROOT), "mtime_utc": datetime.datetime.utcfromtimestamp(os.path.getmtime(path)).strftime("%Y-%m-%dT%H:%M:%SZ"),
            "sha16": hashlib.sha256(raw).hexdigest()[:16], "head": raw[:96].decode("utf-8", "replace").replace("\n", " ")}


def corr_roofline(ops, B=8, N=4096, C=256, iters=10):
    import torch
    f1 = torch.randn(B, N, C, device="cuda")
    f2 = torch.randn(B, N, C, device="cuda")
    vol = torch.empty(B, N, N, device="cuda")
    for _ in range(2):
        ops.corr_volume(f1, f2, vol)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.corr_volume(f1, f2, vol)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    bytes_ = B * (2 * N * C * 4 + N * N * 4)          # 75.50 MB per sample
    flops = B * 2.0 * N * N * C                       # 8.590 GFLOP per sample
    return dict(ms=ms, hbm_gbs=bytes_ / ms / 1e6, hbm_frac=bytes_ / ms / 1e6 / HBM_PEAK_GBS,
                tflops=flops / ms / 1e9, fma_frac=flops / ms / 1e9 / FP32_MFMA_PEAK_TFLOPS, batch=B)


def cpu_baseline_and_parity(model, ops):
    """CPU oracle (port of the reference path) on one 512x512 pair, all host cores; then the HIP path on the same pair with
    the same (seeded) weights: quality difference in the reference's own metric (evaluate.py:44-65)."""
    import torch
    from oracle import adapter as oadapter            # checker + reported CPU baseline only (never the product path)
    from oracle import spec
    from stitch_amd.data import structured_pair
    sd = spec.seeded_state_dict(1234)
    a, b = structured_pair(512, 512, seed=7)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 16)))     # the GPU box grants a 16-core share per GPU
    with torch.no_grad():
        t0 = time.time()
        ref = oadapter.forward_test_eval(sd, a, b)
        dt = time.time() - t0
    base = dict(value=1.0 / dt, unit="pairs/s", cores=torch.get_num_threads(), kind="port",
                sample="1 pair, 512x512, type=test_eval (1 homography + 2 FlowFormer passes), torch-CPU fp32 oracle")
    keep = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.load_state_dict(sd, strict=True)
    try:
        got = model(a.cuda(), b.cuda(), type="test_eval")
        m_hip = ops.masked_psnr_ssim(a.cuda(), got["final_warp_output"])[0].cpu()
        m_ref = ops.masked_psnr_ssim(a.cuda(), ref["final_warp_output"].cuda())[0].cpu()
        dflow = (got["flow_predictions"][0].cpu() - ref["flow_predictions"][0]).abs().flatten()
        flips = int((got["origin_occlusion_mask"].cpu() != ref["origin_occlusion_mask"]).sum())
        parity = {"pair": "structured 512x512 seed 7, seeded weights 1234, HIP vs CPU oracle (reference arithmetic)",
                  "psnr_hip": m_hip[0].item(), "psnr_oracle": m_ref[0].item(), "d_psnr_db": abs(m_hip[0] - m_ref[0]).item(),
                  "d_ssim": abs(m_hip[1] - m_ref[1]).item(), "H_max_abs": (got["H"].cpu() - ref["H"]).abs().max().item(),
                  "flow_max_px": dflow.max().item(), "flow_p99_px": dflow.kthvalue(int(0.99 * dflow.numel())).values.item(),
                  "occlusion_flips": flips, "of_pixels": 512 * 512,
                  "note": "numerical parity of the two implementations, NOT a quality figure: with the seeded random weights (no checkpoint offline) "
                          "the stitched images are ~12.4 dB from image 1 for both paths; the seeded flow network amplifies a ~1e-5 px difference of the "
                          "homography corner offsets ~1e4x, and the CPU oracle run from the HIP path's own offsets moves by the same flow / flip "
                          "amounts (profiles/r5_parity.json: oracle_sensitivity); the enforceable criterion is the stage-held-fixed tests "
                          "(tests/test_parity_gpu.py).  Metric kernel = evaluate.py:44-65 restated from skimage 0.19's published algorithm "
                          "(skimage itself absent: parity vs skimage unpinned)"}
    finally:
        model.load_state_dict(keep, strict=True)
    return base, parity


def write_jpeg_split(root, n, rank=0, world=1):
    """n synthetic 512x512 pairs as UDIS-D lays them out: <root>/testing/input{1,2}/%06d.jpg (quality 95); with several ranks every
    rank writes the files i = rank (mod world)."""
    import numpy as np
    from PIL import Image
    from stitch_amd.data import structured_pair
    for d in ("input1", "input2"):
        os.makedirs(os.path.join(root, "testing", d), exist_ok=True)
    base = [structured_pair(512, 512, seed=40 + i) for i in range(8)]         # 8 distinct scenes, rolled to n distinct pairs
    for i in range(rank, n, world):
        a, b = base[i % 8]
        for d, t in (("input1", a), ("input2", b)):
            arr = np.roll(t[0].permute(1, 2, 0).numpy().astype(np.uint8), (3 * (i // 8), -5 * (i // 8)), (0, 1))
            Image.fromarray(arr).save(os.path.join(root, "testing", d, f"{i:06d}.jpg"), quality=95)


def harness_eval(model, args, rank, world, dist, log):
    """End-to-end throughput of the product evaluation harness (evaluate.py:23-107 -> stitch_amd.evaluate.validate_with_model):
    JPEG files on disk -> (psnr, ssim) table on the host.  Every rank takes its round-robin shard; the clock runs from before the
    dataset is listed to after the final all-gather, max over ranks."""
    import shutil
    import tempfile
    import torch
    from stitch_amd import evaluate as sev
    n = max(world, args.harness_pairs) * (world if world > 1 else 1)
    root = os.path.join(tempfile.gettempdir(), f"stitch_bench_udis_{os.environ.get('MASTER_PORT', 'single')}_{n}")
    if rank == 0:
        shutil.rmtree(root, ignore_errors=True)
    if dist:
        dist.barrier()
    write_jpeg_split(root, n, rank, world)              # (every rank encodes its own share of the files)
    if dist:
        dist.barrier()
    try:
        warm = sev.UDISDataset(root + "/", phase="testing")
        warm.image_list = warm.image_list[:max(2 * args.streams, 6) * world]
        dev = torch.device("cuda", torch.cuda.current_device())
        sev.validate_with_model(model, warm, streams=args.streams, device=dev)       # captures the per-slot hipGraphs, pages the files in
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        ds = sev.UDISDataset(root + "/", phase="testing")
        result, table = sev.validate_with_model(model, ds, streams=args.streams, device=dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], device="cuda")
        if dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # the same pairs through the plain loop (eager launches, one pair at a time, .cpu() per pair): the bits must agree
        sub = sev.UDISDataset(root + "/", phase="testing")
        sub.image_list = sub.image_list[:4 * world]
        t1 = time.perf_counter()
        _, plain = sev.validate_with_model(model, sub, pipelined=False, device=dev)
        dt_plain = time.perf_counter() - t1
        same = bool(torch.equal(plain, table[:len(sub)]))
        if not same:
            log(f"plain loop vs pipelined harness differ:\n{plain}\n{table[:len(sub)]}")
        batched = None
        hb = int(getattr(args, "harness_batch", 0))
        if hb > 1:
            # the same split in batches of hb pairs per forward (evaluate.py:34 batches 12): every GEMM has hb x the rows
            bstreams = max(2, min(args.streams, 3))
            warm.image_list = ds.image_list[:hb * 2 * bstreams * world]
            sev.validate_with_model(model, warm, batch_size=hb, streams=bstreams, device=dev)
            torch.cuda.synchronize()
            if dist:
                dist.barrier()
            t2 = time.perf_counter()
            ds_b = sev.UDISDataset(root + "/", phase="testing")
            _, table_b = sev.validate_with_model(model, ds_b, batch_size=hb, streams=bstreams, device=dev)
            torch.cuda.synchronize()
            tb = torch.tensor([time.perf_counter() - t2], device="cuda")
            if dist:
                dist.all_reduce(tb, op=dist.ReduceOp.MAX)
            sub_b = sev.UDISDataset(root + "/", phase="testing")
            sub_b.image_list = sub_b.image_list[:2 * hb * world]
            _, plain_b = sev.validate_with_model(model, sub_b, batch_size=hb, pipelined=False, device=dev)
            d = (table_b - table).abs()
            batched = {"batch": hb, "streams": bstreams, "pairs_per_s": n / tb.item(), "seconds": tb.item(),
                       "plain_loop_table_equal": bool(torch.equal(plain_b, table_b[:len(sub_b)])),
                       "vs_batch1_psnr_max_abs_dB": float(d[:, 0].max()), "vs_batch1_ssim_max_abs": float(d[:, 1].max())}
        return {"pairs": n, "seconds": tmax.item(), "pairs_per_s": n / tmax.item(), "avg_psnr": result["avg_psnr"], "avg_ssim": result["avg_ssim"],
                "plain_loop_pairs_per_s": len(sub) / dt_plain, "plain_loop_table_equal": same, "batched": batched,
                "what": f"stitch_amd.evaluate.validate_with_model on {n} synthetic 512x512 JPEG pairs on disk (quality 95): PIL decode in "
                        f"{model._eval_pipeline.workers} worker threads, pinned H2D, uint8->float + forward(test_eval) + masked PSNR/SSIM in one "
                        f"hipGraph per pair, {args.streams} pairs in flight, one device->host copy of the table; clock = dataset listing .. "
                        f"table on the host (graphs captured by a {len(warm)}-pair warm-up call)"}
    finally:
        if dist:
            dist.barrier()
        if rank == 0:
            shutil.rmtree(root, ignore_errors=True)


# ---------------------------------------------------------------------------------------------- one rank
def worker(args):
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.dry_run:                                   # launcher + collective plumbing, CPU only
        import torch.distributed as dist
        if world != args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}")
        from stitch_amd import dist as sdist
        dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=sdist.timeout())
        if os.environ.get("ST_BENCH_FAIL_RANK") == str(rank):        # tests/test_dist_cpu.py: one rank dies before the collective
            raise RuntimeError("injected failure (ST_BENCH_FAIL_RANK)")
        vals = torch.full((args.steps,), float(rank))
        gathered = [torch.empty_like(vals) for _ in range(world)]
        dist.all_gather(gathered, vals)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "stitched image-pairs/s at 512x512", "value": 0.0, "unit": "pairs/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                              "gathered_ranks": sorted({int(g[0]) for g in gathered})}), flush=True)
        dist.destroy_process_group()
        return
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s): refusing to print "
                         f"n_gpus={args.gpus} for a {world}-rank run")
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    if "RANK" in os.environ and "MASTER_PORT" in os.environ:      # launched by torch.distributed.run (also with 1 rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from stitch_amd import dist as sdist
        dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=sdist.timeout(),     # "nccl" is RCCL on ROCm
                                device_id=torch.device("cuda", local) if args.backend == "nccl" else None)

    import stitch_amd
    from stitch_amd.data import structured_pair      # deterministic synthetic pairs
    ops = stitch_amd.ops
    cfg, _ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
    torch.manual_seed(1234)
    model = stitch_amd.build_model(cfg).cuda().eval()          # random-init weights of the architecture

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    big = args.workload == "1024"
    if big:
        # configs[3]: 1024x1024 pairs, 4 per GPU, each its own test_out call (batch 1: the canvas is per pair)
        pairs = [tuple(t.cuda() for t in structured_pair(1024, 1024, seed=900 + 8 * rank + i, shift=(11 - 3 * i, 5 * i - 9)))
                 for i in range(4)]
        nb = 1
        nstreams = 1 if args.eager else max(1, args.streams)
        # the network part of test_out (no host sync) replays from a hipGraph; the canvas part stays eager (its shapes depend
        # on the bounds read back to the host).  One graph + HIP stream per pair in flight.
        fwds = [(lambda x, y: model(x, y, type="test_out")) if args.eager else model.graphed_test_out() for _ in range(nstreams)]
        streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nstreams - 1)]

        pending = []

        def finish_one():
            g, h, st = pending.pop(0)
            with torch.cuda.stream(st):
                o = g.finish(h) if g is not None else h
                return o["blend_image"].float().mean().double().reshape(1)     # a per-pair scalar for the final gather

        def step(i=0, n_in_flight=None, drain=False):
            """software pipeline over the pairs in flight: enqueue pair i's network graph, then finish (bounds read-back +
            canvas kernels) the oldest pair whose graph was enqueued n_in_flight - 1 steps ago"""
            depth = n_in_flight or nstreams
            k = i % depth
            a, b = pairs[i % len(pairs)]
            with torch.cuda.stream(streams[k]):
                if args.eager:
                    pending.append((None, model(a, b, type="test_out"), streams[k]))
                else:
                    pending.append((fwds[k], fwds[k].launch(a, b), streams[k]))
            return finish_one() if (len(pending) >= depth or drain) else None
        a, b = pairs[0]
    else:
        ps = [structured_pair(512, 512, seed=7 + rank + 100 * i) for i in range(max(1, args.batch))]
        a, b = torch.cat([p[0] for p in ps]).cuda(), torch.cat([p[1] for p in ps]).cuda()
        nb = a.shape[0]
        nstreams = 1 if args.eager else max(1, args.streams)
        fwds = [(lambda x, y: model(x, y, type="test_eval")) if args.eager else model.graphed("test_eval") for _ in range(nstreams)]
        streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nstreams - 1)]

        def step(i=0, n_in_flight=None):
            """one pair through the hot path on stream i % nstreams (+ its PSNR vs image 1)"""
            k = i % (n_in_flight or nstreams)
            with torch.cuda.stream(streams[k]):
                o = fwds[k](a, b)
                return ops.masked_psnr_ssim(a, o["final_warp_output"])[0]      # evaluate.py:53-59 metric, HIP kernel, fp64 (psnr, ssim)

    t_local_done, rank_times = [0.0], []

    def timed(nsteps, **kw):
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        vals = [step(i, **kw) for i in range(nsteps)]
        if big:
            vals = [v for v in vals if v is not None]
            while pending:
                vals.append(finish_one())
        for st in streams[1:]:
            torch.cuda.current_stream().wait_stream(st)
        metric = torch.stack(vals)
        if dist:
            gathered = [torch.empty_like(metric) for _ in range(world)]
            dist.all_gather(gathered, metric)          # the path's only collective: per-pair metric reduction
        torch.cuda.synchronize()
        t_local_done[0] = time.perf_counter()
        if dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], device="cuda")
        if dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # this rank's own time up to its last kernel (before the barrier): per-rank throughput spread of the N > 1 line
        mine = torch.tensor([t_local_done[0] - t0], device="cuda")
        per_rank = [torch.empty_like(mine) for _ in range(world)] if dist else [mine]
        if dist:
            dist.all_gather(per_rank, mine)
        rank_times[:] = [t.item() for t in per_rank]
        return tmax.item()

    log("model built, warming up")
    # (1) one step per stream captures its hipGraph (two eager forwards + the capture: seconds of host work with the GPU mostly idle);
    # (2) `settle` untimed replays bring the GPU to its steady clocks / power state -- the timed region of a short run (the driver's 20 steps
    # are 0.24 s) otherwise starts on a chip that has been idling through the captures and reads 1-2 % low, more than a round's whole gain;
    # (3) the W warm-up steps of the contract; then exactly K timed steps.
    for i in range(nstreams):
        step(i)
    settle = 0 if args.eager else max(0, 48 // max(1, nb))
    for i in range(settle):
        step(i)
    for i in range(max(args.warmup, nstreams)):
        step(i)
    while big and pending:
        finish_one()
    log("timed region")
    dt = timed(args.steps)
    per_rank_pairs_s = [args.steps * nb / t for t in rank_times]
    dt1 = None
    if nstreams > 1:
        dt1 = timed(max(10, args.steps // 2), n_in_flight=1)       # same graphs, one pair in flight (latency-bound figure)

    harness = None
    if args.harness == "eval" and not big and not args.eager and nb == 1:
        log("product harness (validate_with_model on JPEG pairs)")
        harness = harness_eval(model, args, rank, world, dist, log)

    if rank == 0:
        log(f"timed region done: {dt:.3f} s for {args.steps} steps; instrumented step")
        inst = instrumented_step((lambda: model(a, b, type="test_out")) if big else (lambda: model(a, b, type="test_eval")))
        flops, gemm_ms, launches, abytes = inst["flops"], inst["ms"], inst["launches"], inst["alg_bytes"]
        tf = flops / gemm_ms / 1e9
        split3_on = inst["split3_launches"] > 0
        s3_tf = inst["split3_flops"] / inst["split3_ms"] / 1e9 if split3_on else None
        f32_ms = gemm_ms - inst["split3_ms"]
        value_exact = None
        if split3_on and world == 1 and not big and nb == 1 and not args.eager and os.environ.get("ST_BENCH_CHILD") != "1":
            # the same command on the fp32-MFMA kernels of rounds 1-5 (ST_SPLIT3=0), in a child process (the switch is read at import)
            log("child run with ST_SPLIT3=0 (value_exact_fp32)")
            env = dict(os.environ, ST_SPLIT3="0", ST_BENCH_CHILD="1")
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(args.steps), "--warmup", str(args.warmup), "--streams", str(args.streams),
                   "--harness", "none", "--no-cpu-baseline", "--no-corr-roofline"]
            try:
                cp = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=400)
                for ln in cp.stdout.splitlines():
                    if ln.startswith("{") and '"metric"' in ln:
                        value_exact = json.loads(ln)["value"]
            except Exception as ex:       # the headline does not depend on it
                log(f"value_exact_fp32 child failed: {ex}")
        traffic, tsrc = None, None             # HBM bytes per step of the GEMM family from the committed PMC passes
        for name in ("r6_traffic.json", "r5_traffic.json", "r4_traffic.json", "r3_traffic.json", "r2_traffic.json", "r1_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tpath) and not big and nb == 1:
                t = json.load(open(tpath))
                traffic = t["fetch_bytes_per_step"] + t["write_bytes_per_step"]
                tsrc = dict(provenance(tpath), how="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 corrections applied "
                                                   "(tools/run_pmc_shapes.sh)", launches_per_step_in_file=t.get("launches_per_step"))
                break
        # per-kernel time of the same command from the committed rocprofv3 pass (tools/final_prof.sh -> profiles/r5_kernel_summary.json)
        prof, psrc = None, None
        for name in ("r6_kernel_summary.json", "r5_kernel_summary.json", "r4_kernel_summary.json", "r3_kernel_summary.json"):
            ppath = os.path.join(ROOT, "profiles", name)
            if os.path.exists(ppath) and not big and nb == 1:
                prof = json.load(open(ppath))
                psrc = dict(provenance(ppath), how=prof.get("source"))
                break
        wl = ("synthetic 1024x1024 pairs, 4 per GPU, batch=1, FlowHomoAdpater.forward(type=test_out)" if big else
              f"UDIS-D-shaped 512x512 pairs, batch={nb}, FlowHomoAdpater.forward(type=test_eval)")
        out = {
            "metric": "stitched image-pairs/s at 512x512" if not big else "stitched image-pairs/s at 1024x1024",
            "value": world * args.steps * nb / dt, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (decoder contractions, PatchEmbed's third conv and the C = 128 block tails: exact 3xbf16 split of both operands, 6 products on the bf16 matrix cores, fp32 accumulate -- error vs fp64 "
                      "0.83x the fp32 MFMA chain's; everything else fp32)") if split3_on else "f32",
            "value_exact_fp32": value_exact, "data": "synthetic",
            "effective_warmup_steps": nstreams + settle + max(args.warmup, nstreams),
            "config": {"workload": wl, "pairs_per_step_per_gpu": nb,
                       "launch": "eager" if args.eager else ("hipGraph replay of the network part + eager canvas part" if big else "hipGraph replay"),
                       "pairs_in_flight": nstreams * nb,
                       "parallelism": f"pairs sharded over {world} GPU(s), no data-path collective, one all_gather of per-pair metrics"
                                      + (" [REHEARSAL: all ranks share cuda:0]" if args.share_gpu else "")},
            "settle_steps": settle,
            "harness_pairs_per_s": None if harness is None else harness["pairs_per_s"],
            "harness_batched_pairs_per_s": None if harness is None or not harness.get("batched") else harness["batched"]["pairs_per_s"],
            "harness": harness,
            "value_1_in_flight": None if dt1 is None else world * max(10, args.steps // 2) * nb / dt1,
            "per_rank_pairs_per_s": {"min": min(per_rank_pairs_s), "max": max(per_rank_pairs_s), "ranks": len(per_rank_pairs_s)},
            "roofline": {"bound": "mfma", "kernel": "GEMM family: conv_gemm_split3_kernel / _kpar_kernel / _pair_kernel / _persist_kernel + rowmlp128_split3_kernel (decoder contractions, PatchEmbed c4, block tails: exact 3xbf16 split on the bf16 matrix cores) + the fp32-MFMA kernels conv_gemm_dma_kernel + rowstream_gemm_kernel + rowmlp128_kernel + rowchain128_kernel + conv_gemm_kernel + skinny / narrow variants + split-K reducers (every st_conv_gemm / st_mlp128 / st_linear_chain128 launch of one step)",
                         "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TFLOPS,
                         "frac_note": "fp32-EQUIVALENT FLOPs (2 M N K per launch) of the whole family over its kernel time against the fp32-MFMA peak: the split3 launches execute 6 bf16 products per fp32 product on the 16x faster bf16 pipe, so this fraction is not bounded by 1 for them -- `split3` and `fp32_mfma` below price each part against its own pipe",
                         "split3": None if not split3_on else {
                             "kernel": "conv_gemm_split3_kernel / _kernel64 / _kpar_kernel / _pair_kernel / _persist_kernel (csrc/gemm_split3.h), rowmlp128_split3_kernel (csrc/mlp_split3.h)",
                             "launches_per_step": inst["split3_launches"], "kernel_ms_per_step": inst["split3_ms"], "fp32_equivalent_tflops": s3_tf,
                             "bound": "mfma", "achieved": 6.0 * s3_tf, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s (bf16 products executed: 6 per fp32 product)",
                             "frac": 6.0 * s3_tf / BF16_MFMA_PEAK_TFLOPS,
                             "note": "bounded by what feeds the matrix pipe, not by the pipe: two waves of a SIMD issue one v_mfma_f32_32x32x16_bf16 per 16.8 cycles (tools/probes/mfma_bf16_chain.hip), the GEMM loop sits at 29.5 -- a 64x64 tile needs 1 KB of ds_read_b128 (128 B/clk/CU) and 0.5 KB of LDS-DMA (64 B/clk/CU) per MFMA, one consumer wave issues one per 32 cycles -- at 1.47 GHz with the DMA ring running (profiles/r6_split3_clock.txt, r6_split3_sq_counters.txt); the fused block tail at 96 x ~23 + 4 x (VALU instructions) cycles per chunk: each MFMA holds the SIMD's VALU issue (profiles/r6_mlp_split3_diag.txt)"},
                         "fp32_mfma": {"kernel_ms_per_step": f32_ms, "achieved": (flops - inst["split3_flops"]) / f32_ms / 1e9, "peak": FP32_MFMA_PEAK_TFLOPS,
                                       "frac": (flops - inst["split3_flops"]) / f32_ms / 1e9 / FP32_MFMA_PEAK_TFLOPS, "launches_per_step": launches - inst["split3_launches"]},
                         "traffic": traffic, "traffic_unit": "HBM bytes per step (all launches of the family)", "traffic_source": tsrc,
                         "algorithmic_bytes": abytes, "launches_per_step": launches, "gflop_per_step": flops / 1e9,
                         "kernel_ms_per_step": gemm_ms, "kernel_ms_per_step_uncorrected": inst["ms_raw"],
                         "event_bracket_overhead_us": inst["bracket_overhead_us"],
                         "how": "HIP events around every launch of the family on one extra eager step (one pair in flight), minus the "
                                "median reading of an empty event bracket; `frac` = family FLOPs / that time / peak.  ms_per_step is WALL "
                                "time per pair with pairs_in_flight pairs overlapping on separate streams, so it can be smaller than the "
                                "sum of one pair's kernel durations",
                         # whole path: every kernel, launch gap and non-GEMM kernel included, from the timed region itself
                         "whole_path_frac": (world * args.steps * nb / dt) / world * (flops / nb) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                         "kernel_ms_per_step_rocprof": None if prof is None else prof.get("gemm_family_ms_per_forward"),
                         "frac_rocprof": None if prof is None else flops / 1e9 / prof["gemm_family_ms_per_forward"] / FP32_MFMA_PEAK_TFLOPS,
                         "other_kernels_ms_per_step": None if prof is None else prof.get("other_kernels_ms_per_forward"),
                         "rocprof_source": psrc,
                         "stale_profile_warning": None if (tsrc is None or tsrc.get("launches_per_step_in_file") in (None, launches)) else
                         f"the committed PMC pass saw {tsrc['launches_per_step_in_file']} launches per step, this run {launches}: `traffic` is from an older build"},
            "corr_volume": None if args.no_corr_roofline else corr_roofline(ops),
        }
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline + parity")
            out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(model, ops)
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args)          # parent: starts the N ranks as a child process, relays the JSON line
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import stitch_amd                                  # noqa: F401  (package import for the guard only)
        from stitch_amd import dist as sdist
        with sdist.rank_guard("bench.py worker"):          # a failing rank leaves non-zero BEFORE its peers' next collective
            worker(args)
    else:
        worker(args)


if __name__ == "__main__":
    main()
