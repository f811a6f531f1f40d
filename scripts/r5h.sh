mkdir -p gpurun_out/r5h
timeout -k 10 600 python -m pytest tests/test_baseline_configs_gpu.py -m gpu -x -q -k "cohorts_on_their_own" > gpurun_out/r5h/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5h/tests.log
for i in 1 2; do for cfg in "4 4" "4 1" "2 4" "2 1"; do set -- $cfg
  BETAONE_COHORT_LEAD=$2 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --cohorts $1 > gpurun_out/r5h/drv_k$1_lead$2_$i.log 2>&1 || echo "failed $cfg"
done; done
BETAONE_COHORT_LEAD=16 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --cohorts 4 > gpurun_out/r5h/drv_k4_lead16_1.log 2>&1
timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --cohorts 4 > gpurun_out/r5h/drv100_k4_lead4_1.log 2>&1
BETAONE_COHORT_LEAD=1 timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --cohorts 4 > gpurun_out/r5h/drv100_k4_lead1_1.log 2>&1
for f in gpurun_out/r5h/drv*.log; do grep -h '^{' $f | python -c '
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); c=d["config"]; print(sys.argv[1], c["cohorts"], d["value"], d["ms_per_step"], d["step_ms_min_p50_p90_max"], d["host_fraction"])' $f; done
