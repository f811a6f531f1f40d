cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5x
for pr in 100 640; do timeout -k 10 300 python scripts/cohort_timeline.py 4 256 30 $pr > gpurun_out/r5x/timeline_k4_pr$pr.log 2>&1; tail -4 gpurun_out/r5x/timeline_k4_pr$pr.log; done
mkdir -p gpurun_out/r5x/stats
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5x/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r5x/rocprof_bench.log 2>&1
find gpurun_out/r5x/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r5x/kernel_stats.csv
rm -rf gpurun_out/r5x/stats
head -6 gpurun_out/r5x/kernel_stats.csv | cut -c1-160
