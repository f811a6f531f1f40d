mkdir -p gpurun_out/r5j
timeout -k 10 300 python scripts/step_tail.py 2 200 640 > gpurun_out/r5j/tail_k2.log 2>&1; tail -8 gpurun_out/r5j/tail_k2.log
BETAONE_COHORT_CU_MASK=contiguous timeout -k 10 300 python scripts/step_tail.py 2 200 640 > gpurun_out/r5j/tail_k2_masked.log 2>&1; tail -8 gpurun_out/r5j/tail_k2_masked.log
timeout -k 10 300 python scripts/step_tail.py 4 200 640 > gpurun_out/r5j/tail_k4.log 2>&1; tail -8 gpurun_out/r5j/tail_k4.log
