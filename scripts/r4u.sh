mkdir -p gpurun_out/r4u
for k in 1 2 4; do
timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --opening-steps 0 --cohorts $k > gpurun_out/r4u/k$k.log 2>&1
python - gpurun_out/r4u/k$k.log <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline']
        print(d['config']['cohorts'], d['ms_per_step'], 'tower us', r['avg_launch_us'], r['launch_us_p10_p50_p90'], 'conc', r['concurrency'], 'share', r['share_of_wall_time'], 'b2b', r['back_to_back_us'], 'frac_all', r['frac_all_launches'])
PY
done
