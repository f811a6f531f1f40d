set -x
mkdir -p gpurun_out/r4b
B="python bench.py --warmup 5 --no-cpu-baseline --no-roofline --opening-steps 0 --share-gpu --dist-backend gloo --steps 60"
$B --gpus 2 --games 128 > gpurun_out/r4b/share2x128.log 2>&1 &&
$B --gpus 4 --games 64 > gpurun_out/r4b/share4x64.log 2>&1 &&
$B --gpus 2 --games 256 > gpurun_out/r4b/share2x256.log 2>&1 &&
$B --gpus 3 --games 96 > gpurun_out/r4b/share3x96.log 2>&1
grep -h '^{' gpurun_out/r4b/*.log | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print(d['n_gpus'], d['config']['games_per_gpu'], d['value'], d['ms_per_step'])"
