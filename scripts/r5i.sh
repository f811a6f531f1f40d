mkdir -p gpurun_out/r5i
for k in 4 1; do timeout -k 10 300 python scripts/step_tail.py $k 200 640 > gpurun_out/r5i/tail_k$k.log 2>&1; tail -8 gpurun_out/r5i/tail_k$k.log; done
timeout -k 10 300 python scripts/step_tail.py 4 200 200 > gpurun_out/r5i/tail_k4_pr200.log 2>&1; tail -8 gpurun_out/r5i/tail_k4_pr200.log
