#!/bin/bash
# scripts/final_pass.sh <tag> -- what the driver runs at the end of a round plus the profiles kept under profiles/:
# GPU tests, __graft_entry__.smoke(), the default bench line, a kernel trace of the same workload, the differential soak.
set -u
tag=$1
bash scripts/gpu_round.sh $tag tests || exit 1
timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/${tag}_steps.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_cmd.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/${tag}_steps.log
bash scripts/gpu_round.sh $tag trace
timeout -k 10 400 python scripts/parity_soak.py 3 > gpurun_out/${tag}_soak.log 2>&1; echo "soak rc=$?" | tee -a gpurun_out/${tag}_steps.log
