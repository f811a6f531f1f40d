cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6i; mkdir -p $out; : > $out/pmc_forward64.md
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  p=$(echo $pass | cut -d" " -f1); mkdir -p $out/pmc_$p
  timeout -k 10 400 rocprofv3 --pmc $pass --kernel-trace --kernel-include-regex "bo_k_tower_s|bo_k_heads" --output-format csv -d $out/pmc_$p -- python3 scripts/forward_profile.py tower_split 64 > $out/pmc_$p.log 2>&1 || echo "pass $p failed"
  for c in $pass; do python scripts/pmc_summary.py $out/pmc_$p $c >> $out/pmc_forward64.md 2>&1; done
  rm -rf $out/pmc_$p
done
cat $out/pmc_forward64.md
