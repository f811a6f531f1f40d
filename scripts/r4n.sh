set -x
mkdir -p gpurun_out/r4n
for f in 1 0; do
B1_F32=$f timeout -k 10 300 python scripts/b1_probe.py 15 5 256 1 > gpurun_out/r4n/b1_probe_f32_$f.log 2>&1
echo "f32=$f probe rc=$?" >> gpurun_out/r4n/b1_probe_f32_$f.log
grep -h "precision\|graph replay\|max |b1\|wave \|rc=" gpurun_out/r4n/b1_probe_f32_$f.log
done
timeout -k 10 300 python scripts/b1_probe.py 8 2 128 1 > gpurun_out/r4n/b1_probe_10x128.log 2>&1; grep -h "graph replay" gpurun_out/r4n/b1_probe_10x128.log
timeout -k 10 300 python scripts/b1_probe.py 15 5 256 4 > gpurun_out/r4n/b1_probe_20x256_b4.log 2>&1; grep -h "graph replay" gpurun_out/r4n/b1_probe_20x256_b4.log
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu > gpurun_out/r4n/engine_tests.log 2>&1
tail -3 gpurun_out/r4n/engine_tests.log
timeout -k 10 900 python -m pytest tests/test_baseline_configs_gpu.py tests/test_dropin_gpu.py tests/test_c_abi_gpu.py -x -q -m gpu > gpurun_out/r4n/config_tests.log 2>&1
tail -3 gpurun_out/r4n/config_tests.log
timeout -k 10 300 python tests/uci_latency.py > gpurun_out/r4n/uci_latency.log 2>&1
tail -2 gpurun_out/r4n/uci_latency.log
