export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29535
mkdir -p gpurun_out/r04e_trace
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04e_trace -- python bench.py --force-dist --exchange-every 4 --steps 200 --opening-steps 0 --no-cpu-baseline --no-roofline > gpurun_out/r04e_trace.log 2>&1
python scripts/kernel_percentiles.py gpurun_out/r04e_trace bo_k_ nccl rccl copyBuffer Generic > gpurun_out/r04e_trace_percentiles.md 2>&1
rm -f gpurun_out/r04e_trace/*/*kernel_trace.csv
