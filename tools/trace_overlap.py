"""Concurrency analysis of a rocprofv3 kernel trace (p_kernel_trace.csv) of bench.py with several pairs in flight:
how much of the wall time has 0 / 1 / 2 / 3+ kernels resident, and for each kernel name the time during which it ran ALONE
(nothing from another stream beside it) -- that is the time only the kernel's own efficiency can shorten.
usage: python tools/trace_overlap.py <p_kernel_trace.csv> [skip_fraction]"""
import csv, sys, re, collections
path = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.35
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", ""), int(r["Queue_Id"])))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * skip            # steady state: skip the warm-up / capture part
rows = [r for r in rows if r[0] >= lo and "at::" not in r[2] and "spin" not in r[2]]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for i, (s, e, n, q) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active = set(); last = ev[0][0]
hist = collections.Counter(); alone = collections.Counter(); tot = collections.Counter(); cnt = collections.Counter()
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        hist[min(len(active), 4)] += dt
        if len(active) == 1:
            alone[rows[next(iter(active))][2]] += dt
    if d == 1: active.add(i)
    else: active.discard(i)
    last = t
for s, e, n, q in rows:
    tot[n] += e - s; cnt[n] += 1
wall = t1 - t0
nfw = cnt.get("dlt4_kernel", 1)
print(f"window {wall/1e6:.1f} ms, {nfw} forwards, {wall/1e6/nfw:.3f} ms per forward; queues: {sorted(set(r[3] for r in rows))}")
for k in sorted(hist): print(f"  {k}{'+' if k == 4 else ' '} kernels resident: {100*hist[k]/wall:5.1f} %   ({hist[k]/1e6/nfw:.3f} ms per forward)")
print(f"{'kernel':60} {'ms/fw':>7} {'alone ms/fw':>11} {'alone %':>7}")
for n, v in sorted(tot.items(), key=lambda kv: -alone[kv[0]])[:40]:
    print(f"{n[:60]:60} {v/1e6/nfw:7.3f} {alone[n]/1e6/nfw:11.3f} {100*alone[n]/max(v,1):7.1f}")
