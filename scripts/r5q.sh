set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r5q/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> gpurun_out/r5q/gpu_tests.log; tail -4 gpurun_out/r5q/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5q/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r5q/smoke.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5q/bench_driver_cmd.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
for ln in open('gpurun_out/r5q/bench_driver_cmd.log'):
    if ln.startswith('{'):
        d=json.loads(ln); print(d['value'], d['ms_per_step'], d['config']['cohorts'], d['config']['cohort_cu_masks']); r=d['roofline']; print({k:r[k] for k in r if k not in ('note','timing','traffic_source','basis')}); print(d['cpu_baseline'])
PY
mkdir -p gpurun_out/r5q/stats
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5q/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5q/rocprof_bench.log 2>&1
find gpurun_out/r5q/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r5q/kernel_stats.csv
python scripts/kernel_percentiles.py gpurun_out/r5q/stats bo_k_ copyBuffer > gpurun_out/r5q/trace_percentiles.md 2>&1
rm -rf gpurun_out/r5q/stats
grep -h '^{' gpurun_out/r5q/rocprof_bench.log | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); r=d['roofline']; print('under rocprof:', d['ms_per_step'], r['avg_launch_us'], r['launches_timed'], r['concurrency'])"
head -4 gpurun_out/r5q/kernel_stats.csv | cut -c1-200
