"""UDIS-D evaluation harness (re-statement of the reference's evaluate.py:21-107).

One process per GPU: the test pairs are sharded round-robin (``dist.shard_indices``), each rank runs
``model(image1, image2, type="test_eval")`` and the masked PSNR / SSIM kernel on its pairs, one all-gather
collects the per-pair metrics, and every rank computes the reference's summary (descending sort, slices
[0:331], [331:663], [663:-1], means).  JPEG decoding uses PIL (the reference reads through PIL as well,
core/utils/frame_utils.py); cv2 is not needed on this path."""
from __future__ import annotations

import glob
import os

import numpy as np
import torch

from . import dist as sdist
from . import ops


class UDISDataset:
    """test branch of core/datasets.py:329-389: sorted input1/*.jpg, input2/*.jpg -> float RGB [3,H,W] in 0..255."""

    def __init__(self, data_dir, phase="testing"):
        assert phase == "testing", "only the evaluation split is on the stitching path"
        root = os.path.join(data_dir, phase)
        self.image_list = list(zip(sorted(glob.glob(os.path.join(root, "input1", "*.jpg"))),
                                   sorted(glob.glob(os.path.join(root, "input2", "*.jpg")))))

    def __len__(self):
        return len(self.image_list)

    def __getitem__(self, index):
        from PIL import Image
        out = []
        for path in self.image_list[index]:
            arr = np.array(Image.open(path)).astype(np.uint8)
            if arr.ndim == 2:
                arr = np.tile(arr[..., None], (1, 1, 3))
            out.append(torch.from_numpy(arr[..., :3].copy()).permute(2, 0, 1).float())
        return out[0], out[1], self.image_list[index]


def summarize(psnr_list, ssim_list):
    """evaluate.py:68-106 (sort descending; easy/mid/hard = [0:331], [331:663], [663:-1])."""
    p = sorted((float(x) for x in psnr_list), reverse=True)
    s = sorted((float(x) for x in ssim_list), reverse=True)
    pe, pm, ph = sdist.split_easy_mid_hard(p)
    se, sm, sh = sdist.split_easy_mid_hard(s)
    mean = lambda v: float(np.mean(v)) if len(v) else float("nan")
    return {"avg_psnr": mean(p), "avg_ssim": mean(s), "easy_psnr": mean(pe), "mid_psnr": mean(pm), "hard_psnr": mean(ph),
            "easy_ssim": mean(se), "mid_ssim": mean(sm), "hard_ssim": mean(sh)}


@torch.no_grad()
def validate_with_model(model, val_dataset, batch_size=1, device=None, verbose=False):
    """Sharded evaluate.py:23-107.  Returns (result_dict, table[n_pairs, 2] of per-pair (psnr, ssim))."""
    rank, world, local = sdist.init()
    device = device or torch.device("cuda", local)
    n = len(val_dataset)
    mine = sdist.shard_indices(n, rank, world)
    vals = []
    for start in range(0, len(mine), batch_size):
        idx = mine[start:start + batch_size]
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
