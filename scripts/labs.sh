#!/bin/bash
# scripts/labs.sh -- build and run the stand-alone timing labs of scripts/*.hip on the GPU box; logs under gpurun_out/<tag>_<lab>.log
# usage: bash scripts/labs.sh <tag> <lab> [<lab> ...]     lab = the .hip file's name without the extension
set -u
tag=$1; shift
mkdir -p gpurun_out /tmp/labs
for lab in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w scripts/$lab.hip -o /tmp/labs/$lab || exit 1
  echo "=== $lab"
  timeout -k 10 300 /tmp/labs/$lab > gpurun_out/${tag}_$lab.log 2>&1
  rc=$?
  cat gpurun_out/${tag}_$lab.log
  echo "=== $lab rc=$rc"
  if [ $rc -ge 124 ]; then echo "lab $lab timed out or was killed: stopping"; exit $rc; fi
done
