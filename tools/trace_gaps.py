"""GPU occupancy in time from a rocprofv3 --kernel-trace CSV: over the middle half of the trace (steady state), the
fraction of wall time with >= 1 kernel running, the mean number of kernels in flight, the share of wall time in which
only short (< 20 us) kernels run, and the idle share.   python tools/trace_gaps.py <kernel_trace.csv> [<bench.json of that run>]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo, hi = t0 + (t1 - t0) // 4, t0 + 3 * (t1 - t0) // 4
if len(sys.argv) > 2:      # bench.py's JSON line of the traced run: the timed region (CLOCK_MONOTONIC, the trace's clock)
    import json
    a, b = json.load(open(sys.argv[2]))["timed_region_monotonic_ns"]
    if a < t1 and b > t0:
        lo, hi = a, b
    else:
        print("timed region [%d, %d] does not overlap the trace [%d, %d]: using the middle half" % (a, b, t0, t1))
ev = []
for s, e, name, q in rows:
    s2, e2 = max(s, lo), min(e, hi)
    if e2 > s2:
        small = (e - s) < 20000
        ev.append((s2, 1, small)); ev.append((e2, -1, small))
ev.sort()
busy = only_small = conc = 0
n_all = n_small = 0
prev = lo
for t, d, small in ev:
    dt = t - prev
    if n_all > 0:
        busy += dt; conc += dt * n_all
        if n_all == n_small:
            only_small += dt
    prev = t
    n_all += d
    if small:
        n_small += d
wall = hi - lo
queues = defaultdict(int)
for r in rows:
    queues[r[3]] += 1
print("window ms %.1f  busy %.3f  idle %.3f  only-short-kernels %.3f  mean kernels in flight while busy %.2f  hw queues %d" % (
    wall / 1e6, busy / wall, 1 - busy / wall, only_small / wall, conc / max(1, busy), len(queues)))
# serial sum of kernel time by name inside the window, for comparison with the busy time
tot = defaultdict(float)
for s, e, name, q in rows:
    s2, e2 = max(s, lo), min(e, hi)
    if e2 > s2:
        tot[name.split("(")[0]] += e2 - s2
ssum = sum(tot.values())
print("sum of kernel durations / wall %.3f" % (ssum / wall))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:12]:
    print("  %-70s %.3f of wall" % (k[:70], v / wall))
