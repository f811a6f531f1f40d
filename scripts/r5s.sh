mkdir -p gpurun_out/r5s
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r5s/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> gpurun_out/r5s/gpu_tests.log; tail -4 gpurun_out/r5s/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5s/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r5s/smoke.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5s/bench_driver_cmd.log 2>&1; echo "bench rc=$?"
grep -h '^{' gpurun_out/r5s/bench_driver_cmd.log | python -c '
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); c=d["config"]; print(d["value"], d["ms_per_step"], c["cohorts"], c["cohort_cu_masks"], d["step_ms_min_p50_p90_max"], d["unique_nn_evals_per_sec"]); r=d["roofline"]; print(r["achieved"], r["frac"], r["per_launch"], r["concurrency"]); print(d["cpu_baseline"]["value"])'
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --dist-backend gloo --games 128 --steps 20 --no-cpu-baseline --no-roofline > gpurun_out/r5s/two_ranks_gloo.log 2>&1; echo "2 ranks rc=$?"; grep -h '^{' gpurun_out/r5s/two_ranks_gloo.log | cut -c1-300
