# LDS bank-conflict counters + kernel times of the LeNet step (both launch plans):  gpurun -- 'bash tools/experiments/lenet_lds_pmc.sh'
cd /tmp && export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/tools/forth/lenet_steps.4th
for plan in 1 0; do
  echo "== T4_LAZY_DX0=$plan"
  rm -rf /tmp/kt; T4_LAZY_DX0=$plan timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o lenet -- $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $S > /tmp/kt.log 2>&1
  grep ms_for /tmp/kt.log; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) | head -7 | cut -c1-70,112-150
  T4_LAZY_DX0=$plan $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $S | grep ms_for
  rm -rf /tmp/pl; T4_LAZY_DX0=$plan timeout 300 rocprofv3 --kernel-trace -f csv -d /tmp/pl -o p --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -- $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $S > /tmp/pl.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/pl -name '*counter_collection.csv' | head -1) | grep -A6 "^cs_bwd_b\|^cs_fwd" | head -20
done
