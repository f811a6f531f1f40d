mkdir -p gpurun_out/r6d
timeout -k 10 700 python -m pytest tests/test_engine_gpu.py tests/test_baseline_configs_gpu.py tests/test_c_abi_gpu.py tests/test_records_swap_gpu.py -m gpu -x -q > gpurun_out/r6d/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6d/tests.log
for i in 1 2 3; do for c in 1 0; do
  BETAONE_RESULT_PREFETCH=$c timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r6d/drv100_prefetch${c}_$i.log 2>&1 || echo "failed prefetch=$c"
done; done
for c in 1 0; do BETAONE_RESULT_PREFETCH=$c BO_PLY_PROFILE=1 timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r6d/prof_prefetch$c.log 2>&1; grep "ply profile\] turn" gpurun_out/r6d/prof_prefetch$c.log; done
for f in gpurun_out/r6d/drv*.log; do grep -h '^{' $f | python -c '
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print(sys.argv[1], d["value"], d["ms_per_step"], d["step_ms_min_p50_p90_max"])' $f; done
