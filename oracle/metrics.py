"""CPU oracle for the evaluation metric of evaluate.py:53-65 (test infrastructure).

The reference calls scikit-image 0.19.3 (`peak_signal_noise_ratio`, `structural_similarity(...,
multichannel=True, data_range=255)`), which is not installable here: the functions below restate the published
skimage 0.19 algorithm (7x7 uniform window, K1=.01, K2=.03, sample covariance, crop of (win-1)/2 pixels, mean
over channels).  **Parity against skimage itself is unpinned**; the restatement is what the HIP kernel is
checked against."""
from __future__ import annotations

import numpy as np
from scipy.ndimage import uniform_filter


def masked_uint8_pair(image1, warped, valid):
    """evaluate.py:54-57: uint8 truncation of both images and of the mask mean (only exactly 1.0 survives),
    then uint8 products.  image1/warped [3,H,W] float 0..255, valid [1,H,W] float -> two [H,W,3] uint8."""
    a = np.clip(image1.transpose(1, 2, 0), 0, 255).astype(np.uint8)
    b = np.clip(warped.transpose(1, 2, 0), 0, 255).astype(np.uint8)
    m = np.repeat(valid, 3, 0).transpose(1, 2, 0).astype(np.uint8)
    return a * m, b * m


def psnr(a, b, data_range=255):
    """skimage.metrics.peak_signal_noise_ratio on uint8 inputs (float64 MSE)."""
    err = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 10 * np.log10((data_range ** 2) / err)


def ssim(a, b, data_range=255, win=7, K1=0.01, K2=0.03):
    """skimage.metrics.structural_similarity(multichannel=True) for [H,W,C] uint8 inputs."""
    out = []
    NP = win * win
    cov_norm = NP / (NP - 1)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    pad = (win - 1) // 2
    for c in range(a.shape[2]):
        X, Y = a[..., c].astype(np.float64), b[..., c].astype(np.float64)
        ux, uy = uniform_filter(X, win), uniform_filter(Y, win)
        uxx, uyy, uxy = uniform_filter(X * X, win), uniform_filter(Y * Y, win), uniform_filter(X * Y, win)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        out.append(S[pad:-pad, pad:-pad].mean())
    return float(np.mean(out))


def pair_metrics(image1, warped, valid):
    a, b = masked_uint8_pair(image1, warped, valid)
    return psnr(a, b), ssim(a, b)
