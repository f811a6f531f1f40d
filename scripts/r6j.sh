mkdir -p gpurun_out/r6j
for i in 1 2; do for m in full contiguous off interleaved; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --cohorts 4 --cu-masks $m > gpurun_out/r6j/k4_${m}_$i.log 2>&1 || echo "failed $m"
done; done
for f in gpurun_out/r6j/k4*.log; do grep -h '^{' $f | python -c '
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); c=d["config"]; print(sys.argv[1], c["cohort_cu_masks"], d["value"], d["ms_per_step"], d["step_ms_min_p50_p90_max"])' $f; done
