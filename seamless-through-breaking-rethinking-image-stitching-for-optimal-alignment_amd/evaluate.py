"""UDIS-D evaluation harness (re-statement of the reference's evaluate.py:21-107).

One process per GPU: the test pairs are sharded round-robin (``dist.shard_indices``), each rank runs
``model(image1, image2, type="test_eval")`` and the masked PSNR / SSIM kernel on its pairs, one all-gather
collects the per-pair metrics, and every rank computes the reference's summary (descending sort, slices
[0:331], [331:663], [663:-1], means).  JPEG decoding uses PIL (the reference reads through PIL as well,
core/utils/frame_utils.py); cv2 is not needed on this path.

The default loop is a software pipeline (``EvalPipeline``): JPEG pairs are decoded ahead by a small thread pool (PIL releases
the GIL while it decodes), each of k pairs in flight owns a HIP stream, pinned staging buffers and ONE hipGraph that holds the
whole per-pair work -- uint8 HWC -> float CHW, ``forward(type="test_eval")``, the metric kernels -- and the per-pair (psnr, ssim)
rows stay in a device table until one copy at the end.  The host therefore does per pair: two decodes (off-thread), one memcpy
into pinned memory, two async H2D copies, one graph launch and one 16-byte device copy; it never waits for the GPU inside the
loop (the reference's loop synchronises four times per batch, evaluate.py:38-50).  ``pipelined=False`` keeps the plain
one-pair-at-a-time loop (eager launches, ``.cpu()`` per pair): same kernels, same bits -- ``tests/test_harness_gpu.py``."""
from __future__ import annotations

import glob
import os
from collections import deque
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import dist as sdist
from . import ops


def _decode_rgb8(path):
    """core/datasets.py:381-384: PIL decode -> uint8 [H,W,3] (grey images tiled to 3 channels, alpha dropped)."""
    from PIL import Image
    arr = np.array(Image.open(path)).astype(np.uint8)
    if arr.ndim == 2:
        arr = np.tile(arr[..., None], (1, 1, 3))
    return np.ascontiguousarray(arr[..., :3])


class UDISDataset:
    """test branch of core/datasets.py:329-389: sorted input1/*.jpg, input2/*.jpg -> float RGB [3,H,W] in 0..255."""

    def __init__(self, data_dir, phase="testing"):
        assert phase == "testing", "only the evaluation split is on the stitching path"
        root = os.path.join(data_dir, phase)
        self.image_list = list(zip(sorted(glob.glob(os.path.join(root, "input1", "*.jpg"))),
                                   sorted(glob.glob(os.path.join(root, "input2", "*.jpg")))))

    def __len__(self):
        return len(self.image_list)

    def load_u8(self, index):
        """the pair as the decoder leaves it: two uint8 [H,W,3] arrays (the pipelined harness converts on the GPU)."""
        p1, p2 = self.image_list[index]
        return _decode_rgb8(p1), _decode_rgb8(p2)

    def __getitem__(self, index):
        a, b = self.load_u8(index)
        return (torch.from_numpy(a).permute(2, 0, 1).float(), torch.from_numpy(b).permute(2, 0, 1).float(),
                self.image_list[index])


def summarize(psnr_list, ssim_list):
    """evaluate.py:68-106 (sort descending; easy/mid/hard = [0:331], [331:663], [663:-1])."""
    p = sorted((float(x) for x in psnr_list), reverse=True)
    s = sorted((float(x) for x in ssim_list), reverse=True)
    pe, pm, ph = sdist.split_easy_mid_hard(p)
    se, sm, sh = sdist.split_easy_mid_hard(s)
    mean = lambda v: float(np.mean(v)) if len(v) else float("nan")
    return {"avg_psnr": mean(p), "avg_ssim": mean(s), "easy_psnr": mean(pe), "mid_psnr": mean(pm), "hard_psnr": mean(ph),
            "easy_ssim": mean(se), "mid_ssim": mean(sm), "hard_ssim": mean(sh)}


class _Slot:
    """One pair (or batch of same-shaped pairs) in flight: a HIP stream, pinned staging, a device staging buffer the copy stream
    fills, the static float inputs and the hipGraph of the per-pair work (forward + metric) for one input shape."""

    def __init__(self, model, device, shape, u8):
        self.stream = torch.cuda.Stream(device=device)
        self.h2d_done = torch.cuda.Event()              # copy stream: pinned -> stage finished (pinned buffers reusable)
        self.consumed = torch.cuda.Event()              # slot stream: stage -> static inputs finished (stage reusable)
        self.inflight = deque()                         # completion events of this slot's launches (host throttle)
        B, H, W = shape
        host_shape = (B, H, W, 3) if u8 else (B, 3, H, W)
        dt = torch.uint8 if u8 else torch.float32
        self.u8 = u8
        self.pin = [torch.empty(host_shape, dtype=dt).pin_memory() for _ in range(2)]
        self.pin_np = [p.numpy() for p in self.pin]
        self.stage = [torch.empty(host_shape, dtype=dt, device=device) for _ in range(2)]
        self.inputs = [torch.zeros((B, 3, H, W), dtype=torch.float32, device=device) for _ in range(2)]
        self.metric = torch.empty((B, 2), dtype=torch.float64, device=device)
        ws = ops.new_workspace(device)                 # this graph's own split-K slabs (graphs replay concurrently)

        def body():
            out = model(self.inputs[0], self.inputs[1], type="test_eval")
            ops.masked_psnr_ssim(self.inputs[0], out["final_warp_output"], out=self.metric)        # evaluate.py:44-59

        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), ops.workspace_scope(ws):
            for _ in range(2):                          # warm-up: weight prepack, constant tables, LDS attributes
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), ops.workspace_scope(ws):
            body()
        self._keep = ws

    def unstage(self):
        """device staging -> the graph's static float inputs, on the slot's stream (core/datasets.py:383-386 on the GPU)."""
        for st, dst in zip(self.stage, self.inputs):
            if self.u8:
                ops.load_rgb8(st, out=dst)
            else:
                dst.copy_(st, non_blocking=True)


class EvalPipeline:
    """k pairs in flight, decode-ahead, metric on the device until the end (module docstring).  Kept on the model
    (``model._eval_pipeline``) so that the graphs are captured once per (input shape, slot).

    Per pair the host (1) takes the decoded arrays from the worker pool, (2) copies them into the slot's pinned buffers, (3)
    enqueues the two H2D copies on a COPY stream of its own -- an H2D copy enqueued on the slot's compute stream would sit behind
    the slot's previous graph and the next graph behind the copy: measured 75.6 instead of 81.8 pairs/s
    (a phase-timed copy of this loop, round 4; tools/harness_profile.py sweeps streams / decode threads) -- and (4) on the slot's stream: wait for the copy, uint8 HWC -> float CHW into the graph's static
    inputs, replay the graph, copy the (psnr, ssim) row into the device table.  At most two launches per slot are outstanding
    (the host waits on the completion event of the launch before the previous one), so the GPU always has work queued and the
    host never runs more than 2 k pairs ahead."""

    MAX_OUTSTANDING = 2
    MAX_SHAPES = 4          # distinct (batch, H, W, dtype) kept captured; each costs nslots x (a forward's activations + 64 MiB workspace + pinned staging)

    def __init__(self, model, device, streams=3, decode_workers=None):
        self.model, self.device, self.nslots = model, device, max(1, int(streams))
        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 1
        self.workers = decode_workers or max(2, min(4, ncpu // 2))     # 1.7 ms per 512x512 pair and thread: 4 threads feed ~2 000 pairs/s
        self.copy_stream = torch.cuda.Stream(device=device)
        self._slots = {}                # (shape, u8) -> {slot index: _Slot}, in least-recently-used order
        self._gen = None

    def _slot(self, k, shape, u8):
        """The slot (stream + graph) number k for this input shape; captured on first use.  The cache is bounded: a dataset of many image
        sizes (or ragged batch tails) evicts the least recently used SHAPE -- its graphs, private pools and pinned buffers go with it."""
        key = (shape, u8)
        group = self._slots.pop(key, None)
        if group is None:
            group = {}
            while len(self._slots) >= self.MAX_SHAPES:
                old = self._slots.pop(next(iter(self._slots)))
                for sl in old.values():
                    sl.stream.synchronize()
                    sl.graph = None
                old.clear()
        self._slots[key] = group                        # most recently used last
        if k not in group:
            group[k] = _Slot(self.model, self.device, shape, u8)
        return group[k]

    def n_captured(self):
        return sum(len(g) for g in self._slots.values())

    def _check_weights(self):
        """Graphs replay against the packed weights they were captured with: drop them all when the model's weights changed identity since
        (sub-module ``load_state_dict``, ``.to()`` / ``.float()``, in-place parameter writes) -- ``FlowHomoAdpater.weights_generation``."""
        gen = self.model.weights_generation(deep=True) if hasattr(self.model, "weights_generation") else None
        if gen != self._gen:
            if self._slots:
                torch.cuda.synchronize(self.device)
            self._slots.clear()
            self._gen = gen

    @staticmethod
    def _load(dataset, idx):
        """host side of one work item: decode (worker thread) -> list of (array pair) + whether they are uint8 HWC."""
        if hasattr(dataset, "load_u8"):
            return [dataset.load_u8(i) for i in idx], True
        items = [dataset[i] for i in idx]              # generic dataset: float [3,H,W] tensors, uploaded as they are
        return [(it[0].float().contiguous().numpy(), it[1].float().contiguous().numpy()) for it in items], False

    def _launch(self, slot, sub, table, row):
        while len(slot.inflight) >= self.MAX_OUTSTANDING:
            slot.inflight.popleft().synchronize()
        slot.h2d_done.synchronize()                    # the previous H2D out of this slot's pinned buffers (long done)
        for i, (a, b) in enumerate(sub):
            np.copyto(slot.pin_np[0][i], a)
            np.copyto(slot.pin_np[1][i], b)
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(slot.consumed)         # the stage buffers' previous content has been converted
            slot.stage[0].copy_(slot.pin[0], non_blocking=True)
            slot.stage[1].copy_(slot.pin[1], non_blocking=True)
            slot.h2d_done.record()
        with torch.cuda.stream(slot.stream):
            slot.stream.wait_event(slot.h2d_done)
            slot.unstage()
            slot.consumed.record()
            slot.graph.replay()
            table[row:row + len(sub)].copy_(slot.metric, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            slot.inflight.append(done)

    def run(self, dataset, groups):
        """groups: list of index lists (one forward each).  Returns a CPU fp64 [sum(len(g)), 2] table in group order."""
        self._check_weights()
        n_rows = sum(len(g) for g in groups)
        table = torch.full((max(1, n_rows), 2), float("nan"), dtype=torch.float64, device=self.device)
        ready = torch.cuda.Event()
        ready.record()                                  # the slot streams write rows of `table`: after its fill
        depth = max(2 * self.nslots, self.workers)
        used = set()
        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            futs = deque(pool.submit(self._load, dataset, g) for g in groups[:depth])
            nxt, row, launch = len(futs), 0, 0
            for g in groups:
                pairs, u8 = futs.popleft().result()
                if nxt < len(groups):
                    futs.append(pool.submit(self._load, dataset, groups[nxt]))
                    nxt += 1
                shapes = {p[0].shape for p in pairs} | {p[1].shape for p in pairs}
                # one forward needs one shape: a batch of mixed shapes runs pair by pair (as the plain loop does)
                for sub in ([pairs] if len(shapes) == 1 else [[p] for p in pairs]):
                    if sub[0][0].shape != sub[0][1].shape:
                        raise ValueError(f"image1 {sub[0][0].shape} and image2 {sub[0][1].shape} differ in shape")
                    H, W = (sub[0][0].shape[0:2] if u8 else sub[0][0].shape[1:3])
                    if u8 and (H * W) % 4:
                        sub, u8s = [tuple(np.ascontiguousarray(x.transpose(2, 0, 1), dtype=np.float32) for x in p) for p in sub], False
                    else:
                        u8s = u8
                    slot = self._slot(launch % self.nslots, (len(sub), H, W), u8s)
                    launch += 1
                    if slot not in used:
                        slot.stream.wait_event(ready)
                        used.add(slot)
                    self._launch(slot, sub, table, row)
                    row += len(sub)
        for slot in used:
            torch.cuda.current_stream().wait_stream(slot.stream)
            slot.inflight.clear()
        return table[:n_rows].cpu()                    # the loop's one synchronisation


def _groups(dataset, mine, batch_size):
    """consecutive runs of <= batch_size indices (the reference's DataLoader order, evaluate.py:34); batch_size 1 = one pair each."""
    return [mine[s:s + batch_size] for s in range(0, len(mine), batch_size)]


@torch.no_grad()
def validate_with_model(model, val_dataset, batch_size=1, device=None, verbose=False, pipelined=True, streams=3, decode_workers=None):
    """Sharded evaluate.py:23-107.  Returns (result_dict, table[n_pairs, 2] of per-pair (psnr, ssim))."""
    rank, world, local = sdist.init()
    device = device or torch.device("cuda", local)
    n = len(val_dataset)
    mine = sdist.shard_indices(n, rank, world)
    if model.training:
        raise NotImplementedError("inference-only drop-in: call .eval() first")
    if pipelined:
        pipe = getattr(model, "_eval_pipeline", None)
        if pipe is None or pipe.nslots != max(1, int(streams)) or pipe.device != device:
            pipe = EvalPipeline(model, device, streams=streams, decode_workers=decode_workers)
            model._eval_pipeline = pipe
        with torch.cuda.device(device):
            vals = pipe.run(val_dataset, _groups(val_dataset, mine, batch_size)).tolist() if mine else []
        if verbose:
            for i, (p, s) in zip(mine, vals):
                print(f"rank {rank}: i = {i}, psnr = {p:.6f} ssim = {s:.6f}")
    else:
        vals = []
        for idx in _groups(val_dataset, mine, batch_size):
            items = [val_dataset[i] for i in idx]
            same = len({tuple(it[0].shape) for it in items}) == 1
            groups = [items] if same else [[it] for it in items]
            for grp in groups:
                a = torch.stack([it[0] for it in grp]).to(device)
                b = torch.stack([it[1] for it in grp]).to(device)
                out = model(a, b, type="test_eval")
                m = ops.masked_psnr_ssim(a.contiguous(), out["final_warp_output"]).cpu()
                vals.extend(m.tolist())
                if verbose:
                    for (p, s) in m.tolist():
                        print(f"rank {rank}: psnr = {p:.6f} ssim = {s:.6f}")
    table = sdist.gather_metrics(mine, vals if vals else torch.zeros((0, 2)), n, k=2)
    return summarize(table[:, 0].tolist(), table[:, 1].tolist()), table
