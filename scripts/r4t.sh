set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4t
for ar in 8 4 8 4; do BETAONE_TOWER256_AR=$ar timeout -k 10 300 python scripts/split_tower_check.py 256 2>&1 | grep "15+5x256 | tower_split" | sed "s/^/AR=$ar /"; done | tee gpurun_out/r4t/tower256_ring.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4t/bench_driver_cmd.log 2>&1; python - <<'PY'
import json
for ln in open('gpurun_out/r4t/bench_driver_cmd.log'):
    if ln.startswith('{'):
        d=json.loads(ln); print(d['value'], d['ms_per_step'], d['config']['cohorts']); r=d['roofline']; print({k:r[k] for k in r if k not in ('note','timing','traffic_source','basis')})
PY
mkdir -p gpurun_out/r4t/stats
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4t/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4t/rocprof_bench.log 2>&1
find gpurun_out/r4t/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4t/kernel_stats.csv
python scripts/kernel_percentiles.py gpurun_out/r4t/stats bo_k_ copyBuffer > gpurun_out/r4t/trace_percentiles.md 2>&1
rm -rf gpurun_out/r4t/stats
grep -h '^{' gpurun_out/r4t/rocprof_bench.log | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); r=d['roofline']; print('under rocprof:', d['ms_per_step'], r['avg_launch_us'], r['launches_timed'], r['concurrency'])"
head -3 gpurun_out/r4t/kernel_stats.csv | cut -c1-200
