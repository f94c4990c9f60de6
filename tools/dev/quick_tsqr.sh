# EKF kernel tests, the kernels alone (192 streams per launch, every launch timed), the single 4K stream and three default runs
mkdir -p gpurun_out/$1
timeout -k 10 200 python -m pytest tests/test_gpu_kernels.py -k "ekf" -x -q > gpurun_out/$1/ekf_tests.log 2>&1 || { tail -30 gpurun_out/$1/ekf_tests.log; exit 1; }
tail -1 gpurun_out/$1/ekf_tests.log
MSKF_BENCH_TIMING_PERIOD=1 timeout -k 5 90 python bench.py --no-cpu --steps 6 --warmup 2 --streams 192 --groups 1 --no-pipeline --gram-steps 0 > gpurun_out/$1/solo.json 2> gpurun_out/$1/solo.err || exit 1
python -c "import json;d=json.load(open('gpurun_out/$1/solo.json'));print('solo', round(d['value']), {k:round(x['avg_us']) for k,x in d['kernels'].items()}, d['checks_failed'])"
timeout -k 5 120 python bench.py --config c5 --streams 1 --groups 1 --no-cpu --steps 20 --warmup 5 > gpurun_out/$1/c5_1.json 2> gpurun_out/$1/c5_1.err || exit 1
python -c "import json;d=json.load(open('gpurun_out/$1/c5_1.json'));print('c5 1 stream', round(d['value']), round(d['value_gram_cholesky']), {k:round(x['avg_us']) for k,x in d['kernels'].items()}, d['checks_failed'])"
for rep in 1 2 3; do
  timeout -k 5 90 python bench.py --no-cpu --steps 20 --warmup 5 > gpurun_out/$1/d_$rep.json 2> gpurun_out/$1/d_$rep.err || exit 2
  python -c "import json;d=json.load(open('gpurun_out/$1/d_$rep.json'));print('default rep$rep', round(d['value']), round(d['value_gram_cholesky']), d['host_phases_ms_per_step']['frames_completed_at_close_by_group'], {k:round(x['avg_us']) for k,x in d['kernels'].items() if 'ekf' in k}, d['checks_failed'])"
done
