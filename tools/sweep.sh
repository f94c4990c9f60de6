#!/bin/bash
# throughput sweep; each argument is GROUPSxSTREAMS[xHWQUEUES[xEXTRA_FLAG]]; one line per configuration goes to $1
out=${1:-gpurun_out/sweep.log}
shift
: > "$out"
for cfg in "$@"; do
  IFS=x read -r g s q f <<< "$cfg"
  q=${q:-16}
  echo "G$g S$s Q$q $f $(GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --no-cpu --groups $g --streams $s --steps 40 --warmup 8 ${f:+--$f} 2>/dev/null | cut -c1-230)" >> "$out" || exit 1
done
