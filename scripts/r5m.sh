mkdir -p gpurun_out/r5m
timeout -k 10 300 python scripts/heads_f16_probe.py > gpurun_out/r5m/probe_nt.log 2>&1; cat gpurun_out/r5m/probe_nt.log
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "f16_heads" > gpurun_out/r5m/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5m/tests.log
