mkdir -p gpurun_out/r5o
for h in 0 1; do BETAONE_HEADS_F16=$h timeout -k 10 400 python bench.py --fast --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r5o/fast_h$h.log 2>&1; echo "heads_f16=$h rc=$?"; tail -2 gpurun_out/r5o/fast_h$h.log | cut -c1-400; done
