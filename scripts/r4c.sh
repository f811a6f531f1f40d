set -x
mkdir -p gpurun_out/r4c
timeout -k 10 900 python -m pytest tests/test_records_swap_gpu.py tests/test_multirank_gpu.py -x -q -m gpu > gpurun_out/r4c/new_tests.log 2>&1
echo "new tests rc=$?" >> gpurun_out/r4c/new_tests.log
timeout -k 10 600 python -m pytest tests/test_baseline_configs_gpu.py -x -q -m gpu -k "cohorts or steady_state" > gpurun_out/r4c/cohort_tests.log 2>&1
echo "cohort tests rc=$?" >> gpurun_out/r4c/cohort_tests.log
timeout -k 10 300 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "watched or fused_epilogue" > gpurun_out/r4c/engine_tests.log 2>&1
echo "engine tests rc=$?" >> gpurun_out/r4c/engine_tests.log
B="python bench.py --warmup 5 --no-cpu-baseline --no-roofline --opening-steps 0 --steps 60"
for k in 1 2 4 8; do timeout -k 10 200 $B --cohorts $k > gpurun_out/r4c/cohorts$k.log 2>&1; done
timeout -k 10 200 $B --cohorts 2 --games 512 > gpurun_out/r4c/cohorts2_g512.log 2>&1
timeout -k 10 200 $B --cohorts 4 --games 512 > gpurun_out/r4c/cohorts4_g512.log 2>&1
tail -n 3 gpurun_out/r4c/*tests.log
grep -h '^{' gpurun_out/r4c/cohorts*.log | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print(d['config']['games_per_gpu'], d['config']['cohorts'], d['value'], d['ms_per_step'], d['step_ms_min_p50_p90_max'])"
