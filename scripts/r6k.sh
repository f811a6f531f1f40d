set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6k
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r6k/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> gpurun_out/r6k/gpu_tests.log; tail -4 gpurun_out/r6k/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6k/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r6k/smoke.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6k/bench_driver_cmd.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
for ln in open('gpurun_out/r6k/bench_driver_cmd.log'):
    if ln.startswith('{'):
        d=json.loads(ln); print(d['value'], d['ms_per_step'], d['config']['cohorts'], d['config']['cohort_cu_masks']); r=d['roofline']; print({k:r[k] for k in r if k not in ('note','timing','traffic_source','basis')}); print(d['cpu_baseline'])
PY
