cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5l
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "f16_heads or fused_heads" > gpurun_out/r5l/tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r5l/tests.log
timeout -k 10 300 python scripts/heads_f16_probe.py > gpurun_out/r5l/probe.log 2>&1; cat gpurun_out/r5l/probe.log
mkdir -p gpurun_out/r5l/stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5l/stats -- python3 scripts/heads_f16_probe.py 131072 > gpurun_out/r5l/probe_prof.log 2>&1
find gpurun_out/r5l/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r5l/kernel_stats.csv
rm -rf gpurun_out/r5l/stats
head -12 gpurun_out/r5l/kernel_stats.csv | cut -c1-160
