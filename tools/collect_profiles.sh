#!/bin/bash
# Collect the judged profile set of a round on the GPU box (run through gpurun from the repo root, two calls: a box is
# granted for at most 20 minutes):
#   tools/collect_profiles.sh <out dir under gpurun_out/> lines      bench lines: the driver's command, 60 steps, C3, C5,
#                                                                    TSQR compression, host-resident images, two ranks on one GPU
#   tools/collect_profiles.sh <out dir under gpurun_out/> stats      only the first step of "counters" (kernel stats + occupancy)
#   tools/collect_profiles.sh <out dir under gpurun_out/> counters   the driver's command under rocprofv3 --kernel-trace --stats
#                                                                    and the PMC passes (each counter group in its own run,
#                                                                    python directly after "--")
# The counter summaries (tools/summarize_{pmc,sq,mfma}.py) land in <out dir>/summaries/ and are copied to profiles/rNN_*.json by hand.
set -o pipefail
OUT=gpurun_out/${1:-prof}
WHAT=${2:-lines}
mkdir -p "$OUT"
export TMPDIR=/tmp
SOLO="python bench.py --no-cpu --steps 6 --warmup 2 --streams 192 --groups 1 --no-pipeline --gram-steps 0"     # one 192-stream launch per kernel: the benched launch shape
SOLO3="python bench.py --config c3 --no-cpu --steps 6 --warmup 2 --streams 16 --groups 1 --no-pipeline --gram-steps 0"
if [ "$WHAT" = lines ]; then
  # (most important first: a collection that runs out of box time still has the head of the list)
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2>> "$OUT/log.txt" || exit 1
  echo "driver command done" ; cut -c1-160 "$OUT/bench_driver_cmd.json"
  timeout -k 10 300 python bench.py --no-cpu > "$OUT/bench_60_steps.json" 2>> "$OUT/log.txt" || exit 2
  timeout -k 10 300 python bench.py --no-cpu --steps 20 --warmup 5 --compression auto > "$OUT/bench_gram.json" 2>> "$OUT/log.txt" || exit 3
  # configs[4] is ONE 4K stream per GPU: the single-stream line
  timeout -k 10 300 python bench.py --config c5 --streams 1 --groups 1 --no-cpu --steps 20 --warmup 5 > "$OUT/bench_c5_1stream.json" 2>> "$OUT/log.txt" || exit 6
  echo "c2 variants done"
  timeout -k 10 400 python bench.py --config c3 > "$OUT/bench_c3.json" 2>> "$OUT/log.txt" || exit 5
  timeout -k 10 400 python bench.py --config c5 --unique 8 > "$OUT/bench_c5.json" 2>> "$OUT/log.txt" || exit 6
  echo "c3 c5 done"
  # two ranks (gloo) sharing the box's one GPU at the DEFAULT preset: what the host share costs when ranks multiply
  timeout -k 10 400 python bench.py --gpus 2 --backend gloo --no-cpu --steps 20 --warmup 5 > "$OUT/bench_2rank_gloo_one_gpu.json" 2>> "$OUT/log.txt" || exit 7
  echo "2 ranks done"
  timeout -k 10 300 python bench.py --no-cpu --steps 20 --warmup 5 --host-images > "$OUT/bench_host_images.json" 2>> "$OUT/log.txt" || exit 4
  timeout -k 10 300 python bench.py --config c5 --streams 8 --groups 1 --no-cpu --steps 20 --warmup 5 > "$OUT/bench_c5_8streams.json" 2>> "$OUT/log.txt" || exit 6
  # the staging copies through hipMemcpyAsync (SDMA) as until the middle of round 4: may stall for a minute, never fails the collection
  MSKF_SDMA_COPIES=1 MSKF_WAIT_TIMEOUT_S=100 timeout -k 10 150 python bench.py --no-cpu --steps 20 --warmup 5 > "$OUT/bench_sdma_copies.json" 2>> "$OUT/log.txt" || echo "sdma-copies run did not finish"
  echo "all lines done"
else
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --no-cpu --steps 20 --warmup 5 --gram-steps 0 > "$OUT/bench_under_rocprof.json" 2>> "$OUT/log.txt" || exit 2
  echo "stats done"
  python tools/trace_busy.py "$OUT"/stats/*/*_kernel_trace.csv > "$OUT/trace_occupancy.txt" 2>> "$OUT/log.txt"
  rm -f "$OUT"/stats/*/*_kernel_trace.csv
  [ "$WHAT" = stats ] && { echo "stats only"; exit 0; }
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $SOLO > /dev/null 2>> "$OUT/log.txt" || exit 3
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $SOLO > /dev/null 2>> "$OUT/log.txt" || exit 4
  echo "hbm done"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_sq" -- $SOLO > /dev/null 2>> "$OUT/log.txt" || exit 5
  echo "sq done"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES --output-format csv -d "$OUT/pmc_mfma_c2" -- $SOLO > /dev/null 2>> "$OUT/log.txt" || exit 6
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES --output-format csv -d "$OUT/pmc_mfma_c3" -- $SOLO3 > /dev/null 2>> "$OUT/log.txt" || exit 7
  echo "mfma done"
  rm -f "$OUT"/pmc_*/*/*_kernel_trace.csv
  # the summaries bench.py reads (copy them to profiles/rNN_*.json); they record the launch shape themselves
  mkdir -p "$OUT/summaries"
  python tools/summarize_pmc.py "$OUT"/pmc_fetch/*/*_counter_collection.csv "$OUT"/pmc_write/*/*_counter_collection.csv "$OUT/summaries/pmc_hbm_traffic.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- $SOLO" 192 > /dev/null
  python tools/summarize_sq.py "$OUT"/pmc_sq/*/*_counter_collection.csv "$OUT/summaries/pmc_sq.json" "rocprofv3 --kernel-trace --pmc SQ_* -- $SOLO" 192 > /dev/null
  python tools/summarize_mfma.py "$OUT"/pmc_mfma_c2/*/*_counter_collection.csv "$OUT/summaries/pmc_mfma_c2.json" "rocprofv3 --kernel-trace --pmc (matrix-core counters) -- $SOLO" 192 > /dev/null
  python tools/summarize_mfma.py "$OUT"/pmc_mfma_c3/*/*_counter_collection.csv "$OUT/summaries/pmc_mfma_c3.json" "rocprofv3 --kernel-trace --pmc (matrix-core counters) -- $SOLO3" 16 > /dev/null
  echo "all counters done"
fi
