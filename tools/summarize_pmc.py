#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/rNN_pmc_hbm_traffic.json.

usage: tools/summarize_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> "<command>" [streams per launch]
Values are RAW counter KB per launch (MI355X_MICROARCH.md, HBM section: FETCH_SIZE under-reports wide coalesced
16 B/lane streams by 2x on gfx950; other access widths are uncalibrated, so no factor is applied).
"""
import collections
import csv
import json
import re
import sys


def agg(path, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for row in csv.DictReader(open(path)):
        if row.get("Counter_Name") != counter:
            continue
        m = re.search(r"(k_[a-z_0-9]+)", row["Kernel_Name"])
        if not m:
            continue
        tot[m.group(1)] += float(row["Counter_Value"])
        cnt[m.group(1)] += 1
    return tot, cnt


def main():
    fetch, write, out, cmd = sys.argv[1:5]
    spl = int(sys.argv[5]) if len(sys.argv) > 5 else 96
    ft, fc = agg(fetch, "FETCH_SIZE")
    wt, wc = agg(write, "WRITE_SIZE")
    res = {"command": cmd, "streams_per_launch": spl, "config": "c2",
           "note": "KB per launch, RAW FETCH_SIZE / WRITE_SIZE (separate passes); "
                   "gfx950 FETCH_SIZE halves wide 16 B/lane streams, narrower widths are uncalibrated: no factor applied",
           "kernels": {k: {"launches": fc[k], "fetch_kb_per_launch": round(ft[k] / fc[k], 1),
                           "write_kb_per_launch": round(wt.get(k, 0) / max(wc.get(k, 1), 1), 1)} for k in sorted(ft)}}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(k, v)


if __name__ == "__main__":
    main()
