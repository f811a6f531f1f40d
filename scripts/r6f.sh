mkdir -p gpurun_out/r6f
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --opening-steps 0"
timeout -k 10 300 $B --sims 400 > gpurun_out/r6f/cfg1_default.log 2>&1
timeout -k 10 300 $B --sims 400 --cohorts 2 > gpurun_out/r6f/cfg1_k2.log 2>&1
timeout -k 10 400 $B --games 512 --net 20x256 --net-dtype fp16 > gpurun_out/r6f/cfg4_default.log 2>&1
timeout -k 10 400 $B --games 512 --net 20x256 --net-dtype fp16 --cohorts 2 > gpurun_out/r6f/cfg4_k2.log 2>&1
timeout -k 10 400 $B --games 512 > gpurun_out/r6f/g512_default.log 2>&1
timeout -k 10 400 $B --games 512 --cohorts 2 > gpurun_out/r6f/g512_k2.log 2>&1
timeout -k 10 400 python bench.py --fast --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r6f/fast.log 2>&1
timeout -k 10 300 python tests/uci_latency.py > gpurun_out/r6f/uci_latency.log 2>&1; tail -1 gpurun_out/r6f/uci_latency.log
for f in gpurun_out/r6f/*.log; do grep -h '^{' $f | python -c '
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); c=d["config"]; print(sys.argv[1], c["games_per_gpu"], c["cohorts"], c.get("cohort_cu_masks"), d["value"], d["ms_per_step"])' $f; done
