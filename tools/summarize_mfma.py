#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass with the matrix-core counters into per-kernel MFMA statistics.

usage: tools/summarize_mfma.py <counter_collection.csv> <out.json> "<command>"
Counters: SQ_INSTS_VALU_MFMA_MOPS_F64 (x512 = FP64 matrix flops, rocprofv3's own MfmaFlopsF64 expression),
SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's matrix pipe is busy, summed over SIMDs), SQ_BUSY_CYCLES (cycles the SQ of
a shader engine has work, summed over SEs), SQ_INSTS_MFMA, SQ_WAVES.  Per kernel, averaged over its dispatches:
  tflops            = 512 * MOPS_F64 / dispatch duration (from the same rows' timestamps)
  mfma_busy_pct_simd = MFMA_BUSY_CYCLES / (duration * clock * 1024 SIMDs): share of ALL matrix pipes' time in use
  mfma_busy_pct_of_sq_busy = MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 SEs * 1024): the same against the time the
                       shader engines had waves at all
The clock is taken as 2.4 GHz (MI355X_MICROARCH.md), so the percentages are lower bounds when the chip clocks down."""
import collections
import csv
import json
import re
import sys

CLOCK_HZ = 2.4e9
N_SIMD = 1024
N_SE = 32
FP64_PEAK_TFLOPS = 78.6


def main():
    path, out, cmd = sys.argv[1:4]
    spl = int(sys.argv[4]) if len(sys.argv) > 4 else 192      # streams per launch of the profiled command
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = collections.defaultdict(float)
    seen = set()
    cnt = collections.Counter()
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1) + (m.group(2) or "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[k] += 1
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    res = {"command": cmd, "streams_per_launch": spl, "note": __doc__.split("Counters:")[1].strip(), "fp64_matrix_peak_tflops": FP64_PEAK_TFLOPS, "kernels": {}}
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU_MFMA_MOPS_F64"]):
        if a["SQ_INSTS_MFMA"] <= 0:
            continue
        flops = 512.0 * a["SQ_INSTS_VALU_MFMA_MOPS_F64"]
        t = max(dur[k], 1e-12)
        res["kernels"][k] = {
            "dispatches": cnt[k], "avg_us": round(1e6 * t / cnt[k], 1), "mfma_insts_per_dispatch": round(a["SQ_INSTS_MFMA"] / cnt[k]),
            "fp64_mfma_gflop_per_dispatch": round(flops / cnt[k] / 1e9, 4), "tflops": round(flops / t / 1e12, 3),
            "frac_of_fp64_matrix_peak": round(flops / t / 1e12 / FP64_PEAK_TFLOPS, 4),
            "mfma_busy_pct_simd": round(100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (t * CLOCK_HZ * N_SIMD), 2),
            "mfma_busy_pct_of_sq_busy": round(100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(a["SQ_BUSY_CYCLES"] / N_SE * N_SIMD, 1.0), 2)}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(k, v)


if __name__ == "__main__":
    main()
