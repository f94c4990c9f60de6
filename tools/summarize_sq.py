#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ_* pass (counter_collection.csv) into per-kernel wave statistics.

usage: tools/summarize_sq.py <counter_collection.csv> <out.json> "<command>"
Counters expected: SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_INSTS_VALU.  SQ_*_CYCLES count quad-cycles (MI355X_MICROARCH.md, PMC section), so
cycles per wave = 4 * SQ_WAVE_CYCLES / SQ_WAVES; the percentages are shares of the wave's resident cycles.
"""
import collections
import csv
import json
import re
import sys


def main():
    path, out, cmd = sys.argv[1:4]
    spl = int(sys.argv[4]) if len(sys.argv) > 4 else 192      # streams per launch of the profiled command
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen, cnt = set(), collections.Counter()
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[k] += 1
    res = {"command": cmd, "streams_per_launch": spl, "note": "per kernel, averaged over its dispatches; *_pct are shares of the waves' resident cycles", "kernels": {}}
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
        w, wc = max(a["SQ_WAVES"], 1.0), max(a["SQ_WAVE_CYCLES"], 1.0)
        res["kernels"][k] = {
            "dispatches": cnt[k], "waves_per_dispatch": round(w / cnt[k], 1), "cycles_per_wave": round(4.0 * wc / w),
            "valu_insts_per_wave": round(a["SQ_INSTS_VALU"] / w), "active_any_pct": round(100 * a["SQ_ACTIVE_INST_ANY"] / wc, 1),
            "active_valu_pct": round(100 * a["SQ_ACTIVE_INST_VALU"] / wc, 1), "wait_any_pct": round(100 * a["SQ_WAIT_ANY"] / wc, 1),
            "wait_inst_pct": round(100 * a["SQ_WAIT_INST_ANY"] / wc, 1)}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(k, v)


if __name__ == "__main__":
    main()
