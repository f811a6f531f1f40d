mkdir -p gpurun_out/r4o
timeout -k 10 600 python scripts/b1_determinism.py > gpurun_out/r4o/det.log 2>&1; tail -8 gpurun_out/r4o/det.log
