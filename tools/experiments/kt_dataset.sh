# the dataset-fed LeNet epoch (IDX file -> pinned staging -> GPU normalise -> step) with and without rocprofv3:
#   gpurun -- 'bash tools/experiments/kt_dataset.sh [tools/forth/lenet_dataset_epoch_nohit.4th]'
cd /tmp && export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/${1:-tools/forth/lenet_dataset_epoch.4th}
python $GRAFT_REPO_ROOT/tools/make_synth_mnist.py /tmp/data/MNIST/raw 8192 1024 > /dev/null
$GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $S | grep -i "ms_\|per"
rm -rf /tmp/kd; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kd -o ds -- $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $S > /tmp/kd.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/kd -name "*.db" | head -1) | head -24 | cut -c1-70,112-150
