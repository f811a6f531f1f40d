#!/bin/bash
# scripts/ab.sh -- same-box A/B of bench.py under two environments: bash scripts/ab.sh <tag> "<ENV=.. for A>" "<ENV=.. for B>" <reps> [bench.py args...]
set -u
tag=$1; envA=$2; envB=$3; reps=$4; shift 4
mkdir -p gpurun_out
for r in $(seq 1 $reps); do
  for arm in A B; do
    if [ $arm = A ]; then ev=$envA; else ev=$envB; fi
    timeout -k 10 400 env $ev python bench.py --no-cpu-baseline --no-roofline "$@" > gpurun_out/${tag}_${arm}_$r.log 2>&1
    rc=$?
    echo "$arm ($ev) rep $r rc=$rc: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/${tag}_${arm}_$r.log | tail -1) $(grep -o '"step_ms_min_p50_p90_max": \[[^]]*\]' gpurun_out/${tag}_${arm}_$r.log | tail -1)"
    if [ $rc -ge 124 ]; then echo "timed out: stopping"; exit $rc; fi
  done
done
