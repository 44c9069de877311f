# per-kernel times of the LeNet training step (BASELINE config #3) under rocprofv3:  gpurun -- 'bash tools/experiments/kt_lenet.sh [ENV=VAL ...]'
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; env "$@" rocprofv3 --kernel-trace --stats -d /tmp/kt -o lenet -- $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $GRAFT_REPO_ROOT/tools/forth/lenet_steps.4th > /tmp/kt.log 2>&1
grep ms_for /tmp/kt.log; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) | head -16 | cut -c1-70,112-150
env "$@" $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $GRAFT_REPO_ROOT/tools/forth/lenet_steps.4th | grep ms_for
