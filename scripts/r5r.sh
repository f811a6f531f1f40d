cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5r; mkdir -p $out; : > $out/pmc_heads_f16.md
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  mkdir -p $out/pmc_$pass
  timeout -k 10 400 rocprofv3 --pmc $pass --kernel-trace --kernel-include-regex "bo_k_heads" --output-format csv -d $out/pmc_$pass -- python3 scripts/heads_f16_probe.py 131072 > $out/pmc_$pass.log 2>&1 || echo "pass $pass failed"
  python scripts/pmc_summary.py $out/pmc_$pass $pass >> $out/pmc_heads_f16.md 2>&1
  rm -rf $out/pmc_$pass
done
cat $out/pmc_heads_f16.md
