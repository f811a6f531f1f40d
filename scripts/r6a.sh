mkdir -p gpurun_out/r6a
timeout -k 10 700 python -m pytest tests/test_baseline_configs_gpu.py tests/test_c_abi_gpu.py -m gpu -x -q > gpurun_out/r6a/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r6a/tests.log
for i in 1 2; do timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r6a/drv100_$i.log 2>&1; done
BO_PLY_PROFILE=1 timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r6a/prof.log 2>&1; grep "ply profile\] turn" gpurun_out/r6a/prof.log
for f in gpurun_out/r6a/drv*.log; do grep -h '^{' $f | python -c '
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print(sys.argv[1], d["value"], d["ms_per_step"], d["step_ms_min_p50_p90_max"])' $f; done
