mkdir -p gpurun_out/r5d
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q -k "masked_stream or cohorts_on_their_own" > gpurun_out/r5d/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5d/tests.log
B='bench.py --warmup 5 --no-cpu-baseline --no-roofline --opening-steps 0 --steps 100 --preroll 200'
for cfg in "2 off" "2 contiguous" "4 contiguous" "4 interleaved" "2 off" "4 contiguous"; do set -- $cfg
  timeout -k 10 300 python $B --cohorts $1 --cu-masks $2 > gpurun_out/r5d/k$1_$2_$RANDOM.log 2>&1 || echo "failed $cfg"
done
grep -h '^{' gpurun_out/r5d/k*.log | python -c '
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); c=d["config"]; print(c["cohorts"], c["cohort_cu_masks"], d["value"], d["ms_per_step"], d["step_ms_min_p50_p90_max"])'
