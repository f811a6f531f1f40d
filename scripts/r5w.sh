mkdir -p gpurun_out/r5w
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q > gpurun_out/r5w/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5w/tests.log
timeout -k 10 300 python scripts/cohort_timeline.py 4 256 30 640 > gpurun_out/r5w/timeline_k4.log 2>&1; tail -4 gpurun_out/r5w/timeline_k4.log
for i in 1 2 3; do timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r5w/drv100_$i.log 2>&1; done
for f in gpurun_out/r5w/drv*.log; do grep -h '^{' $f | python -c '
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); c=d["config"]; print(sys.argv[1], c["cohorts"], d["value"], d["ms_per_step"], d["step_ms_min_p50_p90_max"])' $f; done
