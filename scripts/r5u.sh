mkdir -p gpurun_out/r5u
timeout -k 10 400 python bench.py --gpus 1 --steps 3000 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r5u/long_3000.log 2>&1; echo "3000 steps rc=$?"
grep -h '^{' gpurun_out/r5u/long_3000.log | python -c '
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print(d["value"], d["ms_per_step"], d["step_ms_min_p50_p90_max"], d["games_finished_in_timed_region"], d["games_per_hour_measured"], d["mean_plies_of_finished_games"])'
timeout -k 10 500 python scripts/parity_soak.py 3 > gpurun_out/r5u/soak.log 2>&1; echo "soak rc=$?"; tail -3 gpurun_out/r5u/soak.log
