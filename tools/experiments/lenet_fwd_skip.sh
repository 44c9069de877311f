# cs_fwd without its head's W1 phase / exchange / last-stage stores (LAB library; wrong results, timing only):  gpurun -- 'bash tools/experiments/lenet_fwd_skip.sh'
cd /tmp && export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/tools/forth/lenet_steps.4th
LAB=$GRAFT_REPO_ROOT/tensorforth_amd/libt4hip_lab.so
for v in NONE T4K_STACK_LAB_NOW1 T4K_STACK_LAB_NOXCHG T4K_STACK_LAB_NOSTORE; do
  echo "== $v"
  rm -rf /tmp/kt; env $v=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o lenet -- sh -c "LD_PRELOAD=\"\$LD_PRELOAD:$LAB\" exec $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $S" > /tmp/kt.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) | grep "^cs_bwd_b\|^cs_fwd" | cut -c1-20,70-150
done
