mkdir -p gpurun_out/r5b
for k in 1 2 4 8; do timeout -k 10 300 python scripts/cohort_timeline.py $k > gpurun_out/r5b/timeline_k$k.log 2>&1; tail -4 gpurun_out/r5b/timeline_k$k.log; done
