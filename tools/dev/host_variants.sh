mkdir -p gpurun_out/$1
run() {  # name, env...
  name=$1; shift
  for rep in 1 2 3 4; do
    env "$@" timeout -k 5 90 python bench.py --no-cpu --steps 20 --warmup 5 --gram-steps 0 > gpurun_out/$OUT/${name}_$rep.json 2> gpurun_out/$OUT/${name}_$rep.err || exit 2
    python -c "import json;d=json.load(open('gpurun_out/$OUT/${name}_$rep.json'));h=d['host_phases_ms_per_step'];print('$name rep$rep', round(d['value']), h['frames_completed_at_close_by_group'], [round(x) for x in h['stage_ms_per_frame_by_group']['filter']], d['checks_failed'])"
  done
}
OUT=$1
run eht1 MSKF_BENCH_EKF_HOST_THREADS=1
run eht2 MSKF_BENCH_EKF_HOST_THREADS=2
run nap MSKF_WAIT=nap
run nap2 MSKF_WAIT=nap MSKF_BENCH_EKF_HOST_THREADS=2
