"""ctypes binding of libstitch_gfx950.so (the C-ABI declared in include/stitch_gfx950.h).

Argument types are parsed from the header so the binding cannot drift from the declarations.
There is no fallback: if the shared library is missing or a symbol is absent, import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import re

# torch FIRST: it brings its own HIP runtime (torch/lib/libamdhip64.so, loaded with global symbol scope).  Loading
# libstitch_gfx950.so before torch would bind its hip* calls to a SECOND runtime (/opt/rocm's libamdhip64.so.7 from its
# DT_NEEDED entry): two HIP / HSA runtimes in one process, kernels launched through one on memory and streams owned by the other --
# on the GPU box that ends in hipErrorNoDevice at the first launch.  With torch's runtime already global, the library's calls
# resolve to it.
import torch  # noqa: F401,E402

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libstitch_gfx950.so")
HEADER = os.path.join(HERE, "..", "include", "stitch_gfx950.h")

_SCALARS = {"int32_t": C.c_int32, "int64_t": C.c_int64, "float": C.c_float, "int": C.c_int}


class GemmDesc(C.Structure):
    """Mirror of ``st_gemm_desc`` (include/stitch_gfx950.h)."""
    _fields_ = (
        [(n, C.c_void_p) for n in ("a", "w", "c", "bias", "aux0", "aux1", "aux2", "scale_ptr")]
        + [(n, C.c_int32) for n in ("M", "N", "K", "H", "W", "Cin", "ldx", "kh", "kw", "sh", "sw", "ph", "pw", "Ho",
                                    "Wo", "ldw", "ldc", "ld_aux0", "ld_aux1", "ld_aux2", "aux0_row_div", "aux0_row_mod",
                                    "act", "epi")]
        + [("alpha", C.c_float), ("batch", C.c_int32)]
        + [(n, C.c_int64) for n in ("batch_stride_a", "batch_stride_w", "batch_stride_c")]
        + [("tile_cfg", C.c_int32), ("split_k", C.c_int32), ("workspace", C.c_void_p), ("workspace_floats", C.c_int64),
           ("c2", C.c_void_p), ("ldc2", C.c_int32), ("a_bytes", C.c_uint32), ("w_bytes", C.c_uint32), ("reserved0", C.c_int32),
           ("batch_stride_aux1", C.c_int64), ("dh", C.c_int32), ("dw", C.c_int32), ("a_ln", C.c_int32), ("a_ln_eps", C.c_float),
           ("a2", C.c_void_p), ("a2_channels", C.c_int32), ("reserved1", C.c_int32),
           ("c_t", C.c_void_p), ("ld_ct", C.c_int32), ("reserved2", C.c_int32), ("split3", C.c_int32), ("reserved3", C.c_int32)]
        + [(n, C.c_int64) for n in ("a_plane_stride", "w_plane_stride", "a_rows", "w_rows")]
        + [("c_planes", C.c_void_p)] + [(n, C.c_int64) for n in ("c_plane_stride", "c_plane_rows", "c_plane_batch_rows")]
        + [(n, C.c_int32) for n in ("c_plane_col0", "c_plane_row0", "c_no_f32", "reserved4")]
    )


class ChainLayer(C.Structure):
    """Mirror of ``st_chain_layer``."""
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("res_ptr", C.c_void_p), ("ld_res", C.c_int32), ("act", C.c_int32),
                ("ln", C.c_int32), ("ln_eps", C.c_float), ("res", C.c_int32), ("res_layer", C.c_int32)]


class ChainDesc(C.Structure):
    """Mirror of ``st_chain_desc``."""
    _fields_ = [("a", C.c_void_p), ("out", C.c_void_p), ("lda", C.c_int32), ("ldo", C.c_int32), ("M", C.c_int32), ("nlayers", C.c_int32),
                ("layer", ChainLayer * 3)]


class MlpDesc(C.Structure):
    """Mirror of ``st_mlp_desc``."""
    _fields_ = ([(n, C.c_void_p) for n in ("a", "out", "w1", "b1", "w2", "b2", "res")]
                + [(n, C.c_int32) for n in ("lda", "ldo", "ld_res", "M", "hidden", "ln")] + [("ln_eps", C.c_float), ("reserved", C.c_int32)]
                + [(n, C.c_void_p) for n in ("wp", "bp", "res0")] + [("ld_res0", C.c_int32), ("reserved1", C.c_int32)])


def declared_functions(header=HEADER):
    """{name: [ctypes argtypes]} for every ``int st_*(...)`` declaration in the header."""
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\bint\s+(st_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        args = []
        for a in m.group(2).split(","):
            a = " ".join(a.split())
            if a in ("void", ""):
                continue
            if "*" in a:
                args.append(C.c_void_p)
            else:
                args.append(_SCALARS[a.replace("const ", "").split(" ")[0]])
        out[m.group(1)] = args
    return out


class StitchError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python __graft_entry__.py or "
            f"python {os.path.join(HERE, 'build.py')}). There is no CPU / PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in declared_functions().items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"libstitch_gfx950.so does not export {name} (declared in stitch_gfx950.h)") from e
        fn.argtypes = argtypes
        fn.restype = C.c_int
    return lib


lib = _load()


def check(rc, what):
    if rc != 0:
        raise StitchError(f"{what} failed with code {rc}" + (" (ST_EINVAL: rejected argument)" if rc == 1001 else " (hipError_t)"))
