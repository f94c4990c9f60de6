mkdir -p gpurun_out/$1
timeout -k 10 200 python -m pytest tests/test_gpu_kernels.py -k "ekf" -x -q > gpurun_out/$1/ekf_tests.log 2>&1 || { tail -30 gpurun_out/$1/ekf_tests.log; exit 1; }
tail -1 gpurun_out/$1/ekf_tests.log
SOLO="python bench.py --no-cpu --steps 6 --warmup 2 --streams 192 --groups 1 --no-pipeline --gram-steps 0"
for v in 0 1; do
  MSKF_TQ_VARIANT=$v MSKF_BENCH_TIMING_PERIOD=1 timeout -k 5 90 $SOLO > gpurun_out/$1/solo_v$v.json 2> gpurun_out/$1/solo_v$v.err || exit 1
  python -c "import json;d=json.load(open('gpurun_out/$1/solo_v$v.json'));print('solo v$v', round(d['value']), {k:round(x['avg_us']) for k,x in d['kernels'].items()}, d['checks_failed'])"
done
for v in 0 1; do
for rep in 1 2 3 4; do
  MSKF_TQ_VARIANT=$v timeout -k 5 90 python bench.py --no-cpu --steps 20 --warmup 5 --gram-steps 0 > gpurun_out/$1/v${v}_$rep.json 2> gpurun_out/$1/v${v}_$rep.err || exit 2
  python -c "import json;d=json.load(open('gpurun_out/$1/v${v}_$rep.json'));print('v$v rep$rep', round(d['value']), d['host_phases_ms_per_step']['frames_completed_at_close_by_group'], {k:round(x['avg_us']) for k,x in d['kernels'].items()}, d['checks_failed'])"
done
done
