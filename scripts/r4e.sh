set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
B="bench.py --warmup 5 --no-cpu-baseline --no-roofline --opening-steps 0 --steps 100 --preroll 200"
for k in 1 2 4; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4e/trace_k$k -o t -- python3 $B --cohorts $k > gpurun_out/r4e/trace_k$k.log 2>&1
  python scripts/cohort_trace.py gpurun_out/r4e/trace_k$k > gpurun_out/r4e/trace_k$k.md 2>&1
  python scripts/kernel_percentiles.py gpurun_out/r4e/trace_k$k >> gpurun_out/r4e/trace_k$k.md 2>&1
  find gpurun_out/r4e/trace_k$k -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'tail -n 6000 {} | gzip > gpurun_out/r4e/trace_k'$k'_tail.csv.gz; head -1 {} > gpurun_out/r4e/trace_header.csv'
  rm -rf gpurun_out/r4e/trace_k$k
done
cat gpurun_out/r4e/*.md
