"""ctypes view of the plain-C oracle ``oracle/c/geom_oracle.c`` (test infrastructure)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .build_oracle import build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def linspace(start, end, n):
    out = np.empty(n, np.float32)
    lib().orc_linspace(C.c_float(start), C.c_float(end), C.c_int(n), _p(out))
    return out


def homo_warp(U, theta, out_hw, want_idx=True, want_out=True):
    """U [B,C,H,W] f32, theta [B,3,3] -> (out [B,C,oh,ow] f32, idx [B,oh,ow,4] i32 = x0,x1,y0,y1)."""
    U = np.ascontiguousarray(U, np.float32)
    theta = np.ascontiguousarray(theta, np.float32).reshape(-1, 9)
    B, Cc, H, W = U.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    out = np.empty((B, Cc, oh, ow), np.float32) if want_out else None
    idx = np.empty((B, oh, ow, 4), np.int32) if want_idx else None
    lib().orc_homo_warp(_p(U), _p(theta), _p(out), _p(idx, C.c_int32), B, Cc, H, W, oh, ow)
    return out, idx


def range_map(flow):
    flow = np.ascontiguousarray(flow, np.float32)
    B, _, H, W = flow.shape
    out = np.empty((B, 1, H, W), np.float32)
    lib().orc_range_map(_p(flow), _p(out), B, H, W)
    return out


def morph_open(mask, ksz=19):
    mask = np.ascontiguousarray(mask, np.float32)
    B, Cc, H, W = mask.shape
    out = np.empty_like(mask)
    lib().orc_morph_open(_p(mask), _p(out), B * Cc, H, W, ksz)
    return out


def inverse(A):
    """torch.inverse of [B,3,3] or [B,8,8] fp32 in the reference's (MKL) operation order."""
    A = np.ascontiguousarray(A, np.float32)
    n = A.shape[-1]
    out = np.empty_like(A)
    fn = {3: lib().orc_inv3, 8: lib().orc_inv8}[n]
    for a, o in zip(A.reshape(-1, n, n), out.reshape(-1, n, n)):
        fn(_p(a), _p(o))
    return out


def matmul_small(A, Bm):
    A, Bm = np.ascontiguousarray(A, np.float32), np.ascontiguousarray(Bm, np.float32)
    n, m = A.shape[-2:]
    p = Bm.shape[-1]
    out = np.empty(A.shape[:-2] + (n, p), np.float32)
    for a, b, o in zip(A.reshape(-1, n, m), Bm.reshape(-1, m, p), out.reshape(-1, n, p)):
        lib().orc_matmul_small(_p(a), _p(b), _p(o), n, m, p)
    return out


def dlt4(src, dst):
    """tensor_DLT: src, dst [B,4,2] -> H [B,3,3]."""
    src, dst = np.ascontiguousarray(src, np.float32), np.ascontiguousarray(dst, np.float32)
    H = np.empty((src.shape[0], 9), np.float32)
    for s, d, h in zip(src, dst, H):
        lib().orc_dlt4(_p(s), _p(d), _p(h))
    return H.reshape(-1, 3, 3)


def inverse_f64(A):
    """fp64 inverse of [..., n, n] by Gauss-Jordan with partial pivoting in plain C (no LAPACK: deterministic on every host)."""
    A = np.ascontiguousarray(A, np.float64)
    n = A.shape[-1]
    out = np.empty_like(A)
    work = np.empty(2 * n * n, np.float64)
    fn = lib().orc_inv_f64
    fn.restype = C.c_int
    for a, o in zip(A.reshape(-1, n, n), out.reshape(-1, n, n)):
        if fn(_p(a, C.c_double), _p(o, C.c_double), C.c_int(n), _p(work, C.c_double)):
            raise np.linalg.LinAlgError("singular matrix")
    return out
