# same-box A/B of conv-stack lab variants on the LeNet step (LAB library preloaded):  gpurun -- 'bash tools/experiments/lenet_lab_ab.sh "T4K_STACK_LAB_SKIP=0" "T4K_STACK_LAB_SKIP=512"'
cd /tmp
S=$GRAFT_REPO_ROOT/tools/forth/lenet_steps.4th
LAB=$GRAFT_REPO_ROOT/tensorforth_amd/libt4hip_lab.so
for plan in 1 0; do for rep in 1 2 3; do for cfg in "$@"; do
  echo "lazy=$plan $cfg: $(env $cfg T4_LAZY_DX0=$plan LD_PRELOAD=$LAB timeout 120 $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $S | grep -o 'ms_for_2000 [0-9.]*')"
done; done; done
