mkdir -p gpurun_out/r5c
hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_mask_probe scripts/cu_mask_probe.hip 2>/dev/null && timeout -k 10 120 /tmp/cu_mask_probe > gpurun_out/r5c/cu_mask_probe.log 2>&1; cat gpurun_out/r5c/cu_mask_probe.log
for lay in contiguous interleaved; do for k in 2 4 8; do BETAONE_COHORT_CU_MASK=$lay timeout -k 10 300 python scripts/cohort_timeline.py $k > gpurun_out/r5c/timeline_${lay}_k$k.log 2>&1; tail -4 gpurun_out/r5c/timeline_${lay}_k$k.log; done; done
