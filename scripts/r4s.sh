set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4s
timeout -k 10 300 python scripts/split_tower_check.py 256 > gpurun_out/r4s/split_check.md 2>&1; cat gpurun_out/r4s/split_check.md | tail -8
timeout -k 10 400 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "fused_epilogue or tower_output or split_precision" > gpurun_out/r4s/tower_tests.log 2>&1; tail -2 gpurun_out/r4s/tower_tests.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4s/bench_driver_cmd.log 2>&1; tail -c 3000 gpurun_out/r4s/bench_driver_cmd.log
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --games 512 --net 20x256 --net-dtype fp32 --no-cpu-baseline --no-roofline --opening-steps 0 > gpurun_out/r4s/bench_20x256_fp32.log 2>&1; tail -c 600 gpurun_out/r4s/bench_20x256_fp32.log
mkdir -p gpurun_out/r4s/stats
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4s/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4s/rocprof_bench.log 2>&1
find gpurun_out/r4s/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4s/kernel_stats.csv
python scripts/cohort_trace.py gpurun_out/r4s/stats > gpurun_out/r4s/cohort_trace.md 2>&1
python scripts/kernel_percentiles.py gpurun_out/r4s/stats > gpurun_out/r4s/trace_percentiles.md 2>&1
rm -rf gpurun_out/r4s/stats
head -12 gpurun_out/r4s/kernel_stats.csv; head -20 gpurun_out/r4s/cohort_trace.md
