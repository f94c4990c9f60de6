#!/bin/bash
# throughput sweep over (groups, streams); appends one line per configuration to gpurun_out/sweep.log
out=${1:-gpurun_out/sweep.log}
shift
: > "$out"
for cfg in "$@"; do
  g=${cfg%x*}; s=${cfg#*x}
  echo "G$g S$s $(timeout -k 10 200 python bench.py --no-cpu --groups $g --streams $s --steps 40 --warmup 8 2>/dev/null | cut -c1-230)" >> "$out" || exit 1
done
