#!/bin/bash
# scripts/gpu_round.sh -- one gpurun call: GPU tests, then the bench lines; stops at the first step that timed out / was killed.
# usage (on the GPU box): bash scripts/gpu_round.sh <tag> [steps...]   steps: see the case list below (tests bench bench200 steady cfg1 cfg4 g2048 fast fast16k fastsweep pmcfast uci nccl1 dist2 trace pmctower ...)
set -u
tag=$1; shift
out=gpurun_out
mkdir -p $out
run() {  # name, timeout, command...
  local name=$1 to=$2; shift 2
  echo "=== $name: $*" | tee -a $out/${tag}_steps.log
  timeout -k 10 $to "$@" > $out/${tag}_${name}.log 2>&1
  local rc=$?
  echo "=== $name rc=$rc" | tee -a $out/${tag}_steps.log
  tail -n 4 $out/${tag}_${name}.log
  if [ $rc -ge 124 ]; then echo "step $name timed out or was killed: stopping"; exit $rc; fi
  return 0
}
for step in "$@"; do
  case $step in
    tests)   run gpu_tests 1000 python -m pytest tests -m gpu -q -x ;;
    testsall) run gpu_tests 1000 python -m pytest tests -m gpu -q ;;
    newtests) run gpu_newtests 900 python -m pytest tests/test_baseline_configs_gpu.py tests/test_dropin_gpu.py -m gpu -q -s ;;
    bench)   run bench_default 600 python bench.py ;;
    driver)  run bench_driver_cmd 600 python bench.py --gpus 1 --steps 20 --warmup 5 ;;
    smoke)   run smoke 600 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench3)  run bench_a 300 python bench.py --no-cpu-baseline --no-roofline ; run bench_b 300 python bench.py --no-cpu-baseline --no-roofline ; run bench_c 300 python bench.py --no-cpu-baseline --no-roofline ;;
    bench200) run bench_200a 300 python bench.py --steps 200 --no-cpu-baseline --no-roofline ;
              run bench_200b 300 python bench.py --steps 200 --no-cpu-baseline --no-roofline ;;
    softmax) run bench_softmax_engine 300 python bench.py --softmax engine --no-cpu-baseline --no-roofline ;;
    opening) run bench_opening 300 python bench.py --preroll 0 --no-cpu-baseline --no-roofline ;;
    dist2)   run bench_dist2_gloo 400 python bench.py --gpus 2 --share-gpu --dist-backend gloo --games 128 --steps 40 --exchange-every 4 --no-cpu-baseline --no-roofline ;;
    steady)  run bench_steady600 400 python bench.py --steps 600 --no-cpu-baseline --no-roofline --dump-games $out/${tag}_games.json ;;
    cfg1)    run bench_cfg1 300 python bench.py --sims 400 --no-cpu-baseline --no-roofline ;;
    cfg4)    run bench_cfg4 400 python bench.py --games 512 --net 20x256 --net-dtype fp16 --no-cpu-baseline --no-roofline ;;
    g2048)   run bench_g2048 400 python bench.py --games 2048 --no-cpu-baseline --no-roofline ;;
    fast)    run bench_fast 800 python bench.py --fast ;;
    fast16k) run bench_fast_16k 600 python bench.py --fast --games 16384 ;;
    fastprof) run bench_fast_prof 800 env BO_SELECT_PROFILE=1 python bench.py --fast --games 16384 ;;
    fastsweep) run bench_fast_sweep 900 env BO_SELECT_PROFILE=1 python bench.py --fast --select-sweep ;;
    fastsmall) run bench_fast_256x16 500 python bench.py --fast --games 256 --leaves 16 --sims 800 --preroll 48 --steps 6 --warmup 1 --opening-steps 0 --arena-granules-per-expansion 24 ;;
    fasttests) run gpu_fasttests 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -k "fast_mode" ;;
    uci)     run uci_latency 300 python tests/uci_latency.py ;;
    nccl1)   run bench_nccl1 300 env RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python bench.py --force-dist --exchange-every 4 --steps 60 --no-cpu-baseline --no-roofline ;;
    uci_engine) run uci_latency_engine 300 env BO_UCI_SOFTMAX=engine python tests/uci_latency.py ;;
    stepprof) run stepprof 400 python scripts/step_profile.py 640 100 1 ;
              run stepprof_slow 400 python scripts/step_profile.py 640 100 300000 ;;
    trace)   mkdir -p $out/${tag}_trace; run trace 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python bench.py --steps 200 --opening-steps 0 --no-cpu-baseline --no-roofline ;
             python scripts/kernel_percentiles.py $out/${tag}_trace > $out/${tag}_trace_percentiles.md 2>&1 ; rm -f $out/${tag}_trace/*/*_kernel_trace.csv.keep ; tail -n 40 $out/${tag}_trace_percentiles.md ;;
    drvtrace) mkdir -p $out/${tag}_drvtrace; run drvtrace 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_drvtrace -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline ;
             python scripts/kernel_percentiles.py $out/${tag}_drvtrace > $out/${tag}_drvtrace_percentiles.md 2>&1 ; cp $out/${tag}_drvtrace/*/*_kernel_stats.csv $out/${tag}_drvtrace_kernel_stats.csv 2>/dev/null ; rm -rf $out/${tag}_drvtrace ; tail -n 30 $out/${tag}_drvtrace_percentiles.md ;;
    pmcfast) mkdir -p $out/${tag}_pmcfast; run pmcfast 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "bo_k_fw_select" --output-format csv -d $out/${tag}_pmcfast -- python bench.py --fast --preroll 2 --steps 1 --warmup 0 --opening-steps 0 --no-cpu-baseline --no-graph --roofline-steps 32 ;
             python scripts/pmc_summary.py $out/${tag}_pmcfast FETCH_SIZE 32 > $out/${tag}_pmcfast.md 2>&1 ; cat $out/${tag}_pmcfast.md ; rm -rf $out/${tag}_pmcfast ;;
    pmcfastw) mkdir -p $out/${tag}_pmcfastw; run pmcfastw 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "bo_k_fw_select" --output-format csv -d $out/${tag}_pmcfastw -- python bench.py --fast --preroll 2 --steps 1 --warmup 0 --opening-steps 0 --no-cpu-baseline --no-graph --roofline-steps 32 ;
             python scripts/pmc_summary.py $out/${tag}_pmcfastw WRITE_SIZE 32 > $out/${tag}_pmcfastw.md 2>&1 ; cat $out/${tag}_pmcfastw.md ; rm -rf $out/${tag}_pmcfastw ;;
    pmctower) for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
                 p=$(echo $pass | cut -d" " -f1); mkdir -p $out/${tag}_pmctower_$p;
                 run pmctower_$p 400 rocprofv3 --pmc $pass --kernel-trace --kernel-include-regex "bo_k_tower_wg|bo_k_heads" --output-format csv -d $out/${tag}_pmctower_$p -- python scripts/forward_profile.py tower_wg 256 ;
                 for c in $pass; do python scripts/pmc_summary.py $out/${tag}_pmctower_$p $c >> $out/${tag}_pmctower.md 2>&1 ; done ;
                 rm -rf $out/${tag}_pmctower_$p ;
              done ; cat $out/${tag}_pmctower.md ;;
    pmcsplit) for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
                 p=$(echo $pass | cut -d" " -f1); mkdir -p $out/${tag}_pmcsplit_$p;
                 run pmcsplit_$p 400 rocprofv3 --pmc $pass --kernel-trace --kernel-include-regex "bo_k_tower_s|bo_k_heads" --output-format csv -d $out/${tag}_pmcsplit_$p -- python scripts/forward_profile.py tower_split 256 ;
                 for c in $pass; do python scripts/pmc_summary.py $out/${tag}_pmcsplit_$p $c >> $out/${tag}_pmcsplit.md 2>&1 ; done ;
                 rm -rf $out/${tag}_pmcsplit_$p ;
              done ; cat $out/${tag}_pmcsplit.md ;;
    pmcsplit64) rm -f $out/${tag}_pmcsplit64.md $out/${tag}_pmcsplit64.json;
              for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
                 p=$(echo $pass | cut -d" " -f1); mkdir -p $out/${tag}_pmcsplit64_$p;
                 run pmcsplit64_$p 400 rocprofv3 --pmc $pass --kernel-trace --kernel-include-regex "bo_k_tower_s|bo_k_heads" --output-format csv -d $out/${tag}_pmcsplit64_$p -- python scripts/forward_profile.py tower_split 64 ;
                 for c in $pass; do python scripts/pmc_summary.py $out/${tag}_pmcsplit64_$p $c --json $out/${tag}_pmcsplit64.json >> $out/${tag}_pmcsplit64.md 2>&1 ; done ;
                 rm -rf $out/${tag}_pmcsplit64_$p ;
              done ; cat $out/${tag}_pmcsplit64.md ;;
    dist2nccl) run bench_dist2 400 python bench.py --gpus 2 --share-gpu --dist-backend gloo --games 128 --steps 40 --no-cpu-baseline --no-roofline ;;
    ucitrace) mkdir -p $out/${tag}_ucitrace; run ucitrace 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_ucitrace -- python tests/uci_latency.py ;
             python scripts/kernel_percentiles.py $out/${tag}_ucitrace bo_k_ Cijk softmax conv elementwise > $out/${tag}_ucitrace_percentiles.md 2>&1 ; cat $out/${tag}_ucitrace_percentiles.md ;;
    *) echo "unknown step $step"; exit 1 ;;
  esac
done
echo "=== all steps done" | tee -a $out/${tag}_steps.log
