mkdir -p gpurun_out/r4r
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4r/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> gpurun_out/r4r/gpu_tests.log; tail -5 gpurun_out/r4r/gpu_tests.log
timeout -k 10 300 python tests/uci_latency.py > gpurun_out/r4r/uci_latency.log 2>&1; tail -1 gpurun_out/r4r/uci_latency.log
for f in 1 0; do B1_F32=$f timeout -k 10 300 python scripts/b1_probe.py 15 5 256 1 > gpurun_out/r4r/b1_probe_f32_$f.log 2>&1; grep -h "precision\|graph replay\|wave " gpurun_out/r4r/b1_probe_f32_$f.log; done
