cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h/stats
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6h/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6h/rocprof_bench.log 2>&1
find gpurun_out/r6h/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r6h/kernel_stats.csv
python scripts/kernel_percentiles.py gpurun_out/r6h/stats bo_k_ copyBuffer > gpurun_out/r6h/trace_percentiles.md 2>&1
rm -rf gpurun_out/r6h/stats
python scripts/prof_summary.py gpurun_out/r6h/kernel_stats.csv 12 | cut -c1-180
grep -h '^{' gpurun_out/r6h/rocprof_bench.log | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); r=d['roofline']; print('under rocprof:', d['ms_per_step'], r['avg_launch_us'], r['launch_us_p10_p50_p90'], r['launches_timed'], r['concurrency'], r['share_of_wall_time'])"
