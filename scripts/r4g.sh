set -x
mkdir -p gpurun_out/r4g
B="bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --opening-steps 0"
for i in 1 2 3; do
python $B --cohorts 1 > gpurun_out/r4g/k1_$i.log 2>&1
python $B --cohorts 2 > gpurun_out/r4g/k2_$i.log 2>&1
done
python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-roofline --opening-steps 0 --cohorts 2 > gpurun_out/r4g/k2_200.log 2>&1
python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-roofline --opening-steps 0 --cohorts 1 > gpurun_out/r4g/k1_200.log 2>&1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --opening-steps 0 --games 512 --net 20x256 --net-dtype fp16 --cohorts 2 > gpurun_out/r4g/cfg4_k2.log 2>&1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --opening-steps 0 --games 512 --net 20x256 --net-dtype fp16 --cohorts 1 > gpurun_out/r4g/cfg4_k1.log 2>&1
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --opening-steps 0 --sims 400 --cohorts 2 > gpurun_out/r4g/cfg1_k2.log 2>&1
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --opening-steps 0 --sims 400 --cohorts 1 > gpurun_out/r4g/cfg1_k1.log 2>&1
for f in gpurun_out/r4g/*.log; do echo $f; grep -h '^{' $f | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print(d['config']['games_per_gpu'], d['config']['cohorts'], d['steps'], d['value'], d['ms_per_step'], d['step_ms_min_p50_p90_max'], d['games_per_hour_measured'])"; done
