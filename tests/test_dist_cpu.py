"""N > 1 path on CPU: world_size-2 gloo processes shard pairs and all-gather the per-pair metrics."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json, torch
    sys.path.insert(0, %r)
    import stitch_amd
    from stitch_amd import dist as sd
    rank, world, local = sd.init(backend="gloo")
    n = 11
    idx = sd.shard_indices(n, rank, world)
    vals = [[float(i) * 2.0, float(i) + 0.5] for i in idx]          # stand-ins for (psnr, ssim) of pair i
    table = sd.gather_metrics(idx, vals, n)
    assert table.shape == (n, 2)
    assert torch.equal(table[:, 0], torch.arange(n, dtype=torch.float64) * 2.0)
    assert torch.equal(table[:, 1], torch.arange(n, dtype=torch.float64) + 0.5)
    # a short list: n_pairs < world leaves rank 1 with an empty shard; it must still reach the all_gather
    idx1 = sd.shard_indices(1, rank, world)
    t1 = sd.gather_metrics(idx1, [[7.0, 8.0]] if idx1 else torch.zeros((0, 2)), 1, k=2)
    assert t1.shape == (1, 2) and t1.tolist() == [[7.0, 8.0]], t1
    if rank == 0:
        print("GATHER_OK", len(idx), world)
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()
""") % ROOT


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK 6 2" in outs[0]


def test_shard_is_a_partition():
    from stitch_amd import dist as sd
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in sd.shard_indices(1106, r, world))
        assert seen == list(range(1106))
    t = sd.gather_metrics([0, 2], [[1.0], [3.0]], 3)
    assert t[0, 0] == 1.0 and t[2, 0] == 3.0 and t[1, 0] != t[1, 0]


def test_bench_launcher_starts_n_ranks():
    """`bench.py --gpus 2` (not under torchrun) must start 2 ranks itself and relay ONE JSON line with n_gpus = 2;
    rehearsed on CPU with gloo and --dry-run (no model, no GPU): launcher arguments + rendezvous + all_gather."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse_args(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "1024", "--backend", "gloo", "--dry-run"])
    cmd = bench.launcher_command(args, 29544)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--master-addr" in cmd
    assert cmd[cmd.index("--workload") + 1] == "1024" and cmd[cmd.index("--steps") + 1] == "3"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--backend", "gloo", "--dry-run"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["gathered_ranks"] == [0, 1]


def test_out_py_shards_its_pair_list(tmp_path):
    """`torchrun --nproc-per-node 2 out.py ...` (configs[4] names 8 GPUs): every rank takes pairs rank, rank + N, ... of
    demo.txt; rehearsed with --dry-run (no model, no GPU).  The shards are disjoint and cover the list."""
    import ast
    names = [f"pair{i:02d}" for i in range(5)]
    (tmp_path / "demo.txt").write_text("\n".join(names) + "\n")
    seen = {}
    for r in range(2):
        env = dict({k: v for k, v in os.environ.items() if k not in ("MASTER_PORT",)}, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
        p = subprocess.run([sys.executable, os.path.join(ROOT, "out.py"), "--data_root_path", str(tmp_path) + "/", "--dry-run"], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("DRY_RUN")][0]
        seen[r] = ast.literal_eval(line.split("pairs", 1)[1].strip())
    assert seen[0] == names[0::2] and seen[1] == names[1::2]


def test_init_refuses_a_world_size_mismatch(monkeypatch):
    from stitch_amd import dist as sd
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    import pytest
    with pytest.raises(RuntimeError):
        sd.init(backend="gloo", expect_world=8)
    assert sd.init(backend="gloo", expect_world=1) == (0, 1, int(os.environ.get("LOCAL_RANK", "0")))


def test_init_respects_an_existing_process_group(monkeypatch):
    """`dist.init()` inside a process whose caller already initialised torch.distributed (bench.py's ranks) must take that group's
    backend: with `--backend gloo --share-gpu` rank 1 has LOCAL_RANK = 1 on a 1-GPU box, and the one-process-per-GPU check of the
    nccl path would kill it while rank 0 waits in the next collective (the hang of round 4's first 2-rank rehearsal)."""
    import torch
    import torch.distributed as tdist
    from stitch_amd import dist as sdist
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "3")                       # more than the GPUs visible here (0) or on a 1-GPU box
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29533")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    tdist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert sdist.init() == (0, 1, 3)                        # no RuntimeError: the gloo group is CPU-side
    finally:
        tdist.destroy_process_group()
    with pytest.raises(RuntimeError):
        sdist.init()                                            # no group: the nccl default needs one GPU per rank


def test_failing_rank_stops_the_job_under_the_launcher():
    """One rank raising before a collective must end the JOB, not leave its peer waiting (round 4's 2-rank hang): the rank's body
    runs under `dist.rank_guard` (logs the rank, leaves non-zero at once), torchrun then stops the peer, the launcher relays the
    failure.  CPU rehearsal: gloo, --dry-run, the failure injected into rank 1 before the all_gather rank 0 enters."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["ST_BENCH_FAIL_RANK"] = "1"
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--backend", "gloo", "--dry-run"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    took = time.time() - t0
    err = p.stderr.decode() + p.stdout.decode()
    assert p.returncode != 0, err[-2000:]
    assert "rank 1/2 failed in bench.py worker" in err and "injected failure" in err, err[-3000:]
    assert not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{") and '"metric"' in ln]     # no bench line from a broken job
    assert took < 120, took


GUARD_WORKER = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, %r)
    import stitch_amd
    from stitch_amd import dist as sd
    with sd.rank_guard("test body"):
        rank, world, local = sd.init(backend="gloo")
        if rank == 1:
            raise ValueError("rank 1 breaks before the gather")
        sd.gather_metrics([0], [[1.0, 2.0]], 2, k=2)          # rank 0 waits here for a peer that is gone
        print("UNREACHABLE")
""") % ROOT


def test_peer_of_a_dead_rank_times_out_without_a_launcher(tmp_path):
    """No torchrun above the ranks (plain processes): the dead rank leaves non-zero with its rank in the log, and the survivor falls
    out of the collective (peer closed / the finite `dist.timeout()` every process group here is opened with) instead of waiting forever."""
    script = tmp_path / "guard_worker.py"
    script.write_text(GUARD_WORKER)
    import socket
    with socket.socket() as sk:                       # a free port (a busy fixed one would turn this into a false failure or a 200 s timeout)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", ST_DIST_TIMEOUT_S="20")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=200)[0].decode() for p in procs]
    assert procs[1].returncode == 1 and "rank 1/2 failed in test body: ValueError" in outs[1], outs[1][-2000:]
    assert procs[0].returncode != 0 and "UNREACHABLE" not in outs[0] and "rank 0/2 failed in test body" in outs[0], outs[0][-2000:]


def test_every_process_group_gets_a_finite_timeout():
    """grep-level guard: no init_process_group in the product tree without timeout=."""
    import glob
    import re
    files = [os.path.join(ROOT, f) for f in ("bench.py", "out.py", "evaluate.py")] + glob.glob(os.path.join(ROOT, "seamless-*", "*.py"))
    for f in files:
        src = open(f).read()
        for m in re.finditer(r"init_process_group\(", src):
            call = src[m.start():src.index("\n", src.index(")", m.start()) if ")" in src[m.start():] else m.start())]
            depth, end = 0, m.end() - 1
            for i in range(m.end() - 1, len(src)):
                depth += src[i] == "("
                depth -= src[i] == ")"
                if depth == 0:
                    end = i
                    break
            assert "timeout=" in src[m.start():end], (f, src[m.start():end])
