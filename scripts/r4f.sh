set -x
mkdir -p gpurun_out/r4f
B="bench.py --warmup 5 --no-cpu-baseline --no-roofline --opening-steps 0 --steps 100 --preroll 200"
BO_PLY_PROFILE=1 python $B --cohorts 2 > gpurun_out/r4f/k2_prof.log 2>&1
BO_PLY_PROFILE=1 python $B --cohorts 4 > gpurun_out/r4f/k4_prof.log 2>&1
grep -h "ply profile" gpurun_out/r4f/k2_prof.log; grep -h "ply profile" gpurun_out/r4f/k4_prof.log
