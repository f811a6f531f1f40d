mkdir -p gpurun_out/r5t
timeout -k 10 600 python -m pytest tests/test_baseline_configs_gpu.py tests/test_records_swap_gpu.py -m gpu -x -q > gpurun_out/r5t/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5t/tests.log
for i in 1 2 3; do for d in 1 0; do
  BETAONE_COHORT_DEFER_FINISH=$d timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r5t/drv100_defer${d}_$i.log 2>&1 || echo "failed defer=$d"
done; done
for f in gpurun_out/r5t/drv*.log; do grep -h '^{' $f | python -c '
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); c=d["config"]; print(sys.argv[1], c["cohorts"], d["value"], d["ms_per_step"], d["step_ms_min_p50_p90_max"], d["games_finished_in_timed_region"])' $f; done
