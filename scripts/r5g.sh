mkdir -p gpurun_out/r5g
for pr in 100 640; do for k in 4 1; do timeout -k 10 300 python scripts/cohort_timeline.py $k 256 30 $pr > gpurun_out/r5g/timeline_k${k}_pr$pr.log 2>&1; tail -4 gpurun_out/r5g/timeline_k${k}_pr$pr.log; done; done
