"""Per-kernel ISA summary of the package's HIP sources (no GPU needed): VGPRs, spills, vector-memory loads, and how many of those
loads are followed by a full `s_waitcnt vmcnt(0)` before the next load -- a run of load / wait / load / wait is a chain of dependent
memory round trips (hipcc emits it for conditional loads: `cond ? *p : 0`), which a latency-bound kernel cannot afford.
usage: python tools/isa_waits.py [file.hip ...]"""
import os, re, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "seamless-through-breaking-rethinking-image-stitching-for-optimal-alignment_amd", "csrc")
FLAGS = {"geom.hip": ["-ffp-contract=off"], "metrics.hip": ["-ffp-contract=off"], "composition.hip": ["-ffp-contract=off"], "tps_pipeline.hip": ["-ffp-contract=off"]}

def summarize(path):
    name = os.path.basename(path)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", path, "-o", out] + FLAGS.get(name, []),
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
        body = m.group(2)
        g = lambda k: int(re.search(k + r":\s+(\d+)", body).group(1)) if re.search(k + r":\s+(\d+)", body) else -1
        meta[m.group(1)] = (g(r"\.vgpr_count"), g(r"\.vgpr_spill_count"), g(r"\.private_segment_fixed_size"))
    rows = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        kn, body = m.group(1), m.group(2)
        if kn not in meta:
            continue
        loads = waits0 = chain = 0
        last_was_load = False
        for line in body.splitlines():
            t = line.strip()
            if re.match(r"(global_load|buffer_load|flat_load|scratch_load)", t):
                loads += 1; last_was_load = True
            elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                waits0 += 1
                if last_was_load:
                    chain += 1
                last_was_load = False
            elif re.match(r"(v_|ds_|s_barrier)", t):
                pass
        dem = subprocess.run(["c++filt", kn], capture_output=True, text=True).stdout.strip()
        rows.append((name, dem.split("(")[0][:70], *meta[kn], loads, waits0, chain))
    return rows

if __name__ == "__main__":
    files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    print(f"{'file':18} {'kernel':70} vgpr spill scratch loads vmcnt0 load->vmcnt0")
    for f in files:
        for r in summarize(f if os.path.isabs(f) else os.path.join(CSRC, f)):
            print(f"{r[0]:18} {r[1]:70} {r[2]:4} {r[3]:5} {r[4]:7} {r[5]:5} {r[6]:6} {r[7]:6}")
