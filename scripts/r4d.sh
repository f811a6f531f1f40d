set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
B="bench.py --warmup 5 --no-cpu-baseline --no-roofline --opening-steps 0 --steps 100 --preroll 200"
for k in 1 2 4; do
  rocprofv3 --kernel-trace -d gpurun_out/r4d/trace_k$k -o t -- python3 $B --cohorts $k > gpurun_out/r4d/trace_k$k.log 2>&1
  python scripts/cohort_trace.py gpurun_out/r4d/trace_k$k > gpurun_out/r4d/trace_k$k.md 2>&1
  rm -rf gpurun_out/r4d/trace_k$k
done
GPU_MAX_HW_QUEUES=16 python $B --cohorts 4 > gpurun_out/r4d/k4_q16.log 2>&1
GPU_MAX_HW_QUEUES=16 python $B --cohorts 2 > gpurun_out/r4d/k2_q16.log 2>&1
GPU_MAX_HW_QUEUES=24 python $B --cohorts 4 > gpurun_out/r4d/k4_q24.log 2>&1
BO_PLY_PROFILE=1 python $B --cohorts 1 > gpurun_out/r4d/k1_prof.log 2>&1
cat gpurun_out/r4d/*.md
grep -h '^{' gpurun_out/r4d/k*.log | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print(d['config']['games_per_gpu'], d['config']['cohorts'], d['config']['hw_queues']['value'], d['value'], d['ms_per_step'], d['step_ms_min_p50_p90_max'])"
