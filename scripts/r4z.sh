mkdir -p gpurun_out/r4z
for cfg in "255 1" "255 3" "256 2" "252 3" "252 2"; do set -- $cfg
timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --opening-steps 0 --games $1 --cohorts $2 > gpurun_out/r4z/g$1_k$2.log 2>&1
python - gpurun_out/r4z/g$1_k$2.log <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline']
        print(d['config']['games_per_gpu'], d['config']['cohorts'], d['value'], d['ms_per_step'], 'tower us', r['avg_launch_us'], 'conc', r['concurrency'], 'share', r['share_of_wall_time'])
PY
done
