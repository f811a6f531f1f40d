mkdir -p gpurun_out/r5p
timeout -k 10 400 python bench.py --fast --steps 1 --warmup 0 > gpurun_out/r5p/fast_bench_line.log 2>&1; echo "fast bench rc=$?"
grep -h '^{' gpurun_out/r5p/fast_bench_line.log | python -c '
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print(d["value"], d["unit"], d["ms_per_step"], d.get("arena_full_slots")); r=d["roofline"]; print({k:r[k] for k in ("kernel","achieved","frac","traffic") if k in r})'
timeout -k 10 900 python bench.py --fast --steps 12 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r5p/fast_16_plies.log 2>&1; echo "16 plies rc=$?"
grep -h '^{' gpurun_out/r5p/fast_16_plies.log | python -c '
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print(d["value"], d["unit"], d["ms_per_step"], "arena_full_slots", d.get("arena_full_slots"), d["step_ms_min_p50_p90_max"])'
tail -3 gpurun_out/r5p/fast_16_plies.log | cut -c1-300
