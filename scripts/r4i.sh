set -x
mkdir -p gpurun_out/r4i
timeout -k 10 300 python scripts/b1_probe.py 15 5 256 1 > gpurun_out/r4i/b1_probe.log 2>&1
echo "probe rc=$?" >> gpurun_out/r4i/b1_probe.log
tail -12 gpurun_out/r4i/b1_probe.log
grep -q "probe rc=0" gpurun_out/r4i/b1_probe.log && timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "one_launch or route" > gpurun_out/r4i/b1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4i/b1_tests.log
tail -5 gpurun_out/r4i/b1_tests.log
timeout -k 10 600 python -m pytest tests/test_baseline_configs_gpu.py tests/test_dropin_gpu.py -x -q -m gpu -k "config3 or uci or dropin" > gpurun_out/r4i/uci_tests.log 2>&1
echo "uci tests rc=$?" >> gpurun_out/r4i/uci_tests.log
tail -8 gpurun_out/r4i/uci_tests.log
timeout -k 10 300 python tests/uci_latency.py > gpurun_out/r4i/uci_latency.log 2>&1
tail -5 gpurun_out/r4i/uci_latency.log
