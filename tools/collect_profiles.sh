#!/bin/bash
# Collect the judged profile set of a round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <out dir under gpurun_out/>
# default bench line, the same command under rocprofv3 --kernel-trace --stats, the PMC passes (each counter group in
# its own run, python directly after "--"), the C3 / C5 bench lines.  Summaries are made afterwards with
# tools/summarize_{pmc,sq,mfma}.py and tools/trace_busy.py and copied to profiles/.
set -o pipefail
OUT=gpurun_out/${1:-prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
SOLO="python bench.py --no-cpu --steps 6 --warmup 2 --streams 96 --groups 1 --no-pipeline"
SOLO3="python bench.py --config c3 --no-cpu --steps 6 --warmup 2 --streams 16 --groups 1 --no-pipeline"
run() { echo "== $*" >> "$OUT/log.txt"; "$@" >> "$OUT/log.txt" 2>&1; }
timeout -k 10 300 python bench.py > "$OUT/bench_default.json" 2>> "$OUT/log.txt" || exit 1
echo "default done" ; cut -c1-160 "$OUT/bench_default.json"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --no-cpu > "$OUT/bench_under_rocprof.json" 2>> "$OUT/log.txt" || exit 2
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $SOLO > /dev/null 2>> "$OUT/log.txt" || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $SOLO > /dev/null 2>> "$OUT/log.txt" || exit 4
echo "hbm done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_sq" -- $SOLO > /dev/null 2>> "$OUT/log.txt" || exit 5
echo "sq done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES --output-format csv -d "$OUT/pmc_mfma_c2" -- $SOLO > /dev/null 2>> "$OUT/log.txt" || exit 6
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES --output-format csv -d "$OUT/pmc_mfma_c3" -- $SOLO3 > /dev/null 2>> "$OUT/log.txt" || exit 7
echo "mfma done"
timeout -k 10 300 python bench.py --config c3 > "$OUT/bench_c3.json" 2>> "$OUT/log.txt" || exit 8
timeout -k 10 300 python bench.py --config c5 > "$OUT/bench_c5.json" 2>> "$OUT/log.txt" || exit 9
echo "all done"
# the raw kernel trace of the stats run is large: keep only what trace_busy needs
python tools/trace_busy.py "$OUT"/stats/*/*_kernel_trace.csv > "$OUT/trace_occupancy.txt" 2>> "$OUT/log.txt"
rm -f "$OUT"/stats/*/*_kernel_trace.csv "$OUT"/pmc_*/*/*_kernel_trace.csv
