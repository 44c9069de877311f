# cs_bwd_b without one of its phases (LAB library, T4K_STACK_LAB_SKIP bits: 1 dF partial, 2 dX conv, 4 run backward, 8 X window fill, 16 zero fill, 32 dX copy-out):
# kernel time and LDS conflict counters per variant.   gpurun -- 'bash tools/experiments/lenet_lds_skip.sh 0 1 2 4 8 16 32'
cd /tmp && export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/tools/forth/lenet_steps.4th
LAB=$GRAFT_REPO_ROOT/tensorforth_amd/libt4hip_lab.so
for skip in "$@"; do
  echo "== skip $skip"
  export T4K_STACK_LAB_SKIP=$skip
  rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o lenet -- sh -c "LD_PRELOAD=\"\$LD_PRELOAD:$LAB\" exec $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $S" > /tmp/kt.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) | grep "^cs_bwd_b\|^cs_fwd" | cut -c1-20,70-150
  rm -rf /tmp/pl; timeout 300 rocprofv3 --kernel-trace -f csv -d /tmp/pl -o p --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -- sh -c "LD_PRELOAD=\"\$LD_PRELOAD:$LAB\" exec $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $S" > /tmp/pl.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/pl -name '*counter_collection.csv' | head -1) | grep -A3 "^cs_bwd_b" | tr '\n' ' '; echo
done
