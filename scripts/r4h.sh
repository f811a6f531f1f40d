set -x
mkdir -p gpurun_out/r4h
timeout -k 10 300 python scripts/b1_probe.py 15 5 256 1 > gpurun_out/r4h/b1_probe.log 2>&1
echo "probe rc=$?" >> gpurun_out/r4h/b1_probe.log
tail -15 gpurun_out/r4h/b1_probe.log
grep -q "probe rc=0" gpurun_out/r4h/b1_probe.log && timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "one_launch or fused_epilogue or route" > gpurun_out/r4h/b1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4h/b1_tests.log
tail -15 gpurun_out/r4h/b1_tests.log
