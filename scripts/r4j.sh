set -x
mkdir -p gpurun_out/r4j
for m in 0 1 2; do
BETAONE_B1_MODE=$m timeout -k 10 300 python scripts/b1_probe.py 15 5 256 1 > gpurun_out/r4j/b1_probe_mode$m.log 2>&1
echo "mode $m probe rc=$?" >> gpurun_out/r4j/b1_probe_mode$m.log
grep -h "tower    graph\|max |b1\|rc=" gpurun_out/r4j/b1_probe_mode$m.log
BETAONE_B1_MODE=$m timeout -k 10 300 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "one_launch" > gpurun_out/r4j/b1_tests_mode$m.log 2>&1
tail -2 gpurun_out/r4j/b1_tests_mode$m.log
done
