mkdir -p gpurun_out/s2b
for rep in 1 2 3; do
for xo in 0 3 1 4; do
  MSKF_X_ORDER=$xo timeout -k 10 120 python bench.py --no-cpu --steps 20 --warmup 5 > gpurun_out/s2b/xo${xo}_$rep.json 2>> gpurun_out/s2b/log.txt || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/s2b/xo${xo}_$rep.json"))
print("xo${xo} rep$rep", round(d["value"]), round(d["value_gram_cholesky"]), d["host_phases_ms_per_step"]["frames_completed_at_close_by_group"], flush=True)
PY
done
done
