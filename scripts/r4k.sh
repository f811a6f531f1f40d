set -x
mkdir -p gpurun_out/r4k
for f in 1 0; do
B1_F32=$f timeout -k 10 300 python scripts/b1_probe.py 15 5 256 1 > gpurun_out/r4k/b1_probe_f32_$f.log 2>&1
echo "f32=$f probe rc=$?" >> gpurun_out/r4k/b1_probe_f32_$f.log
grep -h "precision\|graph replay\|max |b1\|vs torch\|rc=" gpurun_out/r4k/b1_probe_f32_$f.log
done
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "one_launch or route or fused_epilogue" > gpurun_out/r4k/b1_tests.log 2>&1
tail -3 gpurun_out/r4k/b1_tests.log
timeout -k 10 600 python -m pytest tests/test_baseline_configs_gpu.py tests/test_dropin_gpu.py -x -q -m gpu -k "config3 or uci or dropin" > gpurun_out/r4k/uci_tests.log 2>&1
tail -3 gpurun_out/r4k/uci_tests.log
timeout -k 10 300 python tests/uci_latency.py > gpurun_out/r4k/uci_latency.log 2>&1
tail -2 gpurun_out/r4k/uci_latency.log
