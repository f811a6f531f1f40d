mkdir -p gpurun_out/r5f
for i in 1 2 3; do for k in 2 4; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --cohorts $k > gpurun_out/r5f/drv_k${k}_$i.log 2>&1 || echo "failed k=$k"
done; done
for k in 2 4; do timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --cohorts $k > gpurun_out/r5f/drv_noroof_k${k}.log 2>&1; done
for f in gpurun_out/r5f/drv_*.log; do grep -h '^{' $f | python -c '
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); c=d["config"]; print(sys.argv[1], c["cohorts"], c["cohort_cu_masks"], d["value"], d["ms_per_step"], d["step_ms_min_p50_p90_max"], d["unique_nn_evals_per_sec"])' $f; done
