mkdir -p gpurun_out/r4w
timeout -k 10 400 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4w/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r4w/smoke.log; tail -8 gpurun_out/r4w/smoke.log
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4w/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> gpurun_out/r4w/gpu_tests.log; tail -5 gpurun_out/r4w/gpu_tests.log
timeout -k 10 600 python bench.py --fast > gpurun_out/r4w/bench_fast.log 2>&1; echo "fast rc=$?" >> gpurun_out/r4w/bench_fast.log; tail -c 1500 gpurun_out/r4w/bench_fast.log
