mkdir -p gpurun_out/r4v
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/fetch_lab.hip -o /tmp/fetch_lab && timeout -k 10 120 /tmp/fetch_lab > gpurun_out/r4v/fetch_lab.log 2>&1; cat gpurun_out/r4v/fetch_lab.log
