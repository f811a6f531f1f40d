mkdir -p gpurun_out/r6l
for i in 1 2 3; do for pr in 640 1600 3200; do
  timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --preroll $pr > gpurun_out/r6l/pr${pr}_$i.log 2>&1 || echo "failed $pr"
done; done
for f in gpurun_out/r6l/pr*.log; do grep -h '^{' $f | python -c '
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); c=d["config"]; print(c["preroll_plies"], d["value"], d["ms_per_step"], d["step_ms_min_p50_p90_max"], d["untimed_setup_seconds"])'; done | sort -n
