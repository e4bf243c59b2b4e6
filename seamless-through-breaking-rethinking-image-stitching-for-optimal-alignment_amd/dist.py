"""Multi-GPU harness of the stitching path: one process per GPU, pairs sharded round-robin, no data-path
collective; one all-gather (RCCL over xGMI on the GPU box, ``nccl`` backend) of the per-pair metrics at
the end.  Replaces the reference's single-process ``nn.DataParallel`` (out.py:80, evaluate.py:119)."""
from __future__ import annotations

import contextlib
import datetime
import os
import sys
import traceback

import torch
import torch.distributed as dist


def timeout(seconds=None):
    """Finite timeout of every process group this package (and bench.py / out.py) opens: ``ST_DIST_TIMEOUT_S`` or 600 s.
    torch's defaults are 10 min (nccl) / 30 min (gloo); what matters is that it is FINITE and passed explicitly, so that a
    rank waiting in a collective for a peer that died does not sit there for the rest of the job."""
    if seconds is None:
        seconds = float(os.environ.get("ST_DIST_TIMEOUT_S", "600"))
    return datetime.timedelta(seconds=float(seconds))


@contextlib.contextmanager
def rank_guard(what="rank body"):
    """Wrap one rank's body: an exception is logged WITH ITS RANK and the process leaves at once with a non-zero code
    (``os._exit``: no destructors, no further collective) so that torchrun stops the peers instead of letting them wait in
    their next collective (round 4's 2-rank hang: rank 1 raised before an all_gather rank 0 had already entered).  Peers not
    under torchrun fall out of the collective after ``timeout()``."""
    try:
        yield
    except BaseException as e:                      # SystemExit(0) passes through; everything else is fatal for the job
        if isinstance(e, SystemExit) and not e.code:
            raise
        rank, world = os.environ.get("RANK", "0"), os.environ.get("WORLD_SIZE", "1")
        sys.stderr.write(f"[stitch_amd.dist] rank {rank}/{world} failed in {what}: {type(e).__name__}: {e}\n")
        traceback.print_exc()
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(e.code if isinstance(e, SystemExit) and isinstance(e.code, int) else 1)


def init(backend=None, expect_world=None, timeout_s=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, world_size, local_rank).

    With a GPU backend the process is bound to ITS GPU here (``torch.cuda.set_device(local_rank)`` and ``device_id=`` for the
    process group), not left to the caller: RCCL then opens its communicator on the right device eagerly and a stray
    ``cuda:0`` default cannot put every rank on one GPU.  ``expect_world``: fail loudly when the launcher started a
    different number of ranks than the caller asked for (a silent 1-rank run must not report N GPUs)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if expect_world is not None and int(expect_world) != world:
        raise RuntimeError(f"launched with WORLD_SIZE={world} but {expect_world} ranks were requested")
    if dist.is_initialized():
        backend = dist.get_backend()         # the caller's process group decides (bench.py --backend gloo --share-gpu: every rank on cuda:0)
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    gpu = backend == "nccl"
    if gpu:
        if local >= torch.cuda.device_count():
            raise RuntimeError(f"LOCAL_RANK={local} but only {torch.cuda.device_count()} GPU(s) are visible: one process per GPU")
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout(timeout_s),
                                device_id=torch.device("cuda", local) if gpu else None)
    if dist.is_initialized() and dist.get_world_size() != world:
        raise RuntimeError(f"process group has {dist.get_world_size()} ranks, environment says {world}")
    return rank, world, local


def shard_indices(n_pairs, rank, world):
    """pair i goes to rank i % world (pairs are independent: no cross-sample op on the path)."""
    return list(range(rank, n_pairs, world))


def gather_metrics(local_idx, local_vals, n_pairs, device=None, k=None):
    """All ranks contribute (index, values[k]) rows; every rank returns the full [n_pairs, k] table.

    The only collective of the path (~16 B per pair): one all_gather of fixed-size padded blocks.  ``k`` (values
    per pair) must be passed whenever a rank can own zero pairs (n_pairs < world): an empty shard cannot infer it,
    and every rank has to reach the all_gather with the same block shape."""
    vals = torch.as_tensor(local_vals, dtype=torch.float64)
    if k is None:
        if len(local_idx) == 0:
            raise ValueError("gather_metrics: pass k= (values per pair) when the local shard may be empty")
        k = vals.reshape(len(local_idx), -1).shape[1]
    vals = vals.reshape(len(local_idx), k)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = torch.full((n_pairs, k), float("nan"), dtype=torch.float64)
        if len(local_idx):
            out[torch.as_tensor(local_idx)] = vals
        return out
    world = dist.get_world_size()
    cap = (n_pairs + world - 1) // world
    dev = device or (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))
    block = torch.full((cap, k + 1), float("nan"), dtype=torch.float64, device=dev)
    if len(local_idx):
        block[:len(local_idx), 0] = torch.as_tensor(local_idx, dtype=torch.float64, device=dev)
        block[:len(local_idx), 1:] = vals.to(dev)
    blocks = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(blocks, block)
    out = torch.full((n_pairs, k), float("nan"), dtype=torch.float64)
    for b in blocks:
        b = b.cpu()
        ok = ~torch.isnan(b[:, 0])
        out[b[ok, 0].long()] = b[ok, 1:]
    return out


def split_easy_mid_hard(psnr_desc_sorted):
    """evaluate.py:76-93: descending sort, slices [0:331], [331:663], [663:-1] (the worst pair is dropped)."""
    s = psnr_desc_sorted
    return s[0:331], s[331:663], s[663:-1]
