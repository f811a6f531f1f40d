mkdir -p gpurun_out/r4p
for m in 0 1 2; do
BETAONE_B1_MODE=$m timeout -k 10 300 python scripts/resume_probe.py > gpurun_out/r4p/mode$m.log 2>&1; echo "mode $m split"; grep -v "aborted" gpurun_out/r4p/mode$m.log | grep -c differs
BETAONE_F32_TOWER=fp32 BETAONE_B1_MODE=$m timeout -k 10 300 python scripts/resume_probe.py > gpurun_out/r4p/mode${m}_f32.log 2>&1; echo "mode $m f32"; grep -v "aborted" gpurun_out/r4p/mode${m}_f32.log | grep -c differs
done
