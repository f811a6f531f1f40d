mkdir -p gpurun_out/r4x
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "fast" > gpurun_out/r4x/fast_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4x/fast_tests.log; tail -4 gpurun_out/r4x/fast_tests.log
grep -q "rc=0" gpurun_out/r4x/fast_tests.log && BO_SELECT_PROFILE=1 timeout -k 10 900 python bench.py --fast --select-sweep > gpurun_out/r4x/sweep.log 2> gpurun_out/r4x/sweep.err; grep "sweep\|select profile" gpurun_out/r4x/sweep.err | tail -20; tail -c 1200 gpurun_out/r4x/sweep.log
