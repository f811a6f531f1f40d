cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "fast or heads or fused" > gpurun_out/r5n/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r5n/tests.log
timeout -k 10 400 python bench.py --fast --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5n/fast_bench.log 2>&1; echo "fast bench rc=$?"
grep -h '^{' gpurun_out/r5n/fast_bench.log | python -c '
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print(d["value"], d["unit"], d["ms_per_step"], d.get("unique_nn_evals_per_sec")); r=d["roofline"]; print({k:r[k] for k in ("kernel","achieved","frac") if k in r})'
mkdir -p gpurun_out/r5n/stats
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5n/stats -- python3 bench.py --fast --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r5n/rocprof_fast.log 2>&1
find gpurun_out/r5n/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r5n/fast_kernel_stats.csv
rm -rf gpurun_out/r5n/stats
head -14 gpurun_out/r5n/fast_kernel_stats.csv | cut -c1-150
