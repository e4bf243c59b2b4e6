"""Build libstitch_gfx950.so (hipcc, gfx950 only) in-tree: ``python <pkg>/build.py``."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libstitch_gfx950.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# geom.hip pins every fp32 rounding (bit-exact sample indices): no fma contraction there
SOURCES = {"gemm.hip": [], "nn.hip": [], "flowops.hip": [], "geom.hip": ["-ffp-contract=off"],
           "metrics.hip": ["-ffp-contract=off"], "operators.hip": [], "composition.hip": ["-ffp-contract=off"], "tps_pipeline.hip": ["-ffp-contract=off"], "patchembed.hip": []}
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# No packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) anywhere in the library.  Measured in round 6
# (tools/neighbour_stress.py, profiles/r6_pk_f32_beside_bf16_mfma.txt): while another wave of the SIMD issues v_mfma_f32_32x32x16_bf16
# back to back (the split3 kernels, csrc/gemm_split3.h), a wave's v_pk_*_f32 results come back wrong in lanes 48-63 -- 80 % of the
# launches of the bilinear-resize kernel beside a split3 GEMM had corrupted 16-lane runs (DMA-only variant of the aggressor: none;
# MFMA-only: the same; the fp32-MFMA kernels as aggressor: none; torch's own elementwise kernels as victims: none).  The compiler's
# hazard tables cannot see another wave, so the instructions are simply not generated (scalar v_mul / v_add / v_fma instead: same
# IEEE results, and beside an MFMA stream the packed forms were the slower choice anyway -- MI355X_MICROARCH.md, packed f32 VALU).
COMMON += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
COMMON += [f for f in os.environ.get("ST_EXP_FLAGS", "").split() if f]      # experiment switches (e.g. -DST_EXP_PRIO), never set for the shipped build
if os.environ.get("ST_EXACT_TRANSCENDENTALS", "0") == "1":       # diagnostic build (csrc/common.h): not the shipped arithmetic
    COMMON.append("-DST_EXACT_TRANSCENDENTALS")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_split3.h"), os.path.join(CSRC, "mlp_split3.h"), os.path.join(HERE, "..", "include", "stitch_gfx950.h")]
    objs, jobs = [], []
    os.makedirs(os.path.join(CSRC, "_obj"), exist_ok=True)
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, "_obj", src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + COMMON + extra + ["-c", s, "-o", o])
    if jobs:
        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            # (the HOST pass of a .hip file does not know the device-only target feature and says so once per pass: not a diagnostic of ours)
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            err = "".join(l for l in r.stderr.splitlines(True) if "'-packed-fp32-ops' is not a recognized feature for this target" not in l)
            if err:
                sys.stderr.write(err)
            if r.returncode:
                raise subprocess.CalledProcessError(r.returncode, cmd)
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    if jobs or _stale(OUT, objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
