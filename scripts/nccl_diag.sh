export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
MASTER_PORT=29541 python bench.py --force-dist --exchange-every 1000000 --steps 60 --no-cpu-baseline --no-roofline > gpurun_out/diag_nccl_noex.log 2>&1 &&
MASTER_PORT=29542 python bench.py --force-dist --dist-backend gloo --exchange-every 4 --steps 60 --no-cpu-baseline --no-roofline > gpurun_out/diag_gloo1.log 2>&1 &&
MASTER_PORT=29543 python bench.py --force-dist --exchange-every 4 --steps 60 --no-cpu-baseline --no-roofline > gpurun_out/diag_nccl_e4.log 2>&1
