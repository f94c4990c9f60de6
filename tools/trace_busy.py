#!/usr/bin/env python3
"""GPU occupancy of a rocprofv3 --kernel-trace run: share of wall time with >= 1 kernel in flight, mean number of
kernels in flight, and per-kernel time inside a window of the trace.

usage: tools/trace_busy.py <kernel_trace.csv> [window_start_fraction=0.7] [window_end_fraction=1.0]
The window is a fraction of the span between the first and the last kernel (the bench's timed steps are the tail)."""
import collections
import csv
import re
import sys


def main():
    path = sys.argv[1]
    f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
    f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    rows = []
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40]))
    t_lo, t_hi = min(r[0] for r in rows), max(r[1] for r in rows)
    a, b = t_lo + f0 * (t_hi - t_lo), t_lo + f1 * (t_hi - t_lo)
    sel = [(max(s, a), min(e, b), k) for s, e, k in rows if e > a and s < b]
    ev = []
    for s, e, _ in sel:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy, area, depth, last = 0.0, 0.0, 0, a
    for t, dlt in ev:
        if depth > 0:
            busy += t - last
        area += depth * (t - last)
        depth += dlt
        last = t
    span = b - a
    print("window %.1f ms: >=1 kernel in flight %.1f %%, mean kernels in flight %.2f" % (span / 1e6, 100 * busy / span, area / span))
    per = collections.defaultdict(lambda: [0.0, 0])
    for s, e, k in sel:
        per[k][0] += e - s; per[k][1] += 1
    tot = sum(v[0] for v in per.values())
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][0]):
        print("%-28s %8.1f ms %5.1f %% of kernel time  %6d launches  avg %8.1f us" % (k, v[0] / 1e6, 100 * v[0] / tot, v[1], v[0] / v[1] / 1e3))


if __name__ == "__main__":
    main()
