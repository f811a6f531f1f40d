set -x
mkdir -p gpurun_out/r4a
B="python bench.py --warmup 5 --no-cpu-baseline --no-roofline --opening-steps 0"
$B --steps 60 > gpurun_out/r4a/burst1_cfg2.log 2>&1 &&
BETAONE_BURST_TWO_PATHS=0 $B --steps 60 > gpurun_out/r4a/burst0_cfg2.log 2>&1 &&
$B --steps 60 --games 128 > gpurun_out/r4a/g128.log 2>&1 &&
$B --steps 60 --games 512 > gpurun_out/r4a/g512.log 2>&1 &&
$B --steps 30 --games 512 --net 20x256 --net-dtype fp16 > gpurun_out/r4a/burst1_cfg4.log 2>&1 &&
BETAONE_BURST_TWO_PATHS=0 $B --steps 30 --games 512 --net 20x256 --net-dtype fp16 > gpurun_out/r4a/burst0_cfg4.log 2>&1 &&
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4a/driver_cmd.log 2>&1
tail -n 2 gpurun_out/r4a/*.log
