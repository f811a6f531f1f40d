mkdir -p gpurun_out/r5k
for k in 4 2; do BO_PLY_PROFILE=1 timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --cohorts $k > gpurun_out/r5k/prof_k$k.log 2>&1; grep -v "^{" gpurun_out/r5k/prof_k$k.log | tail -22; done
