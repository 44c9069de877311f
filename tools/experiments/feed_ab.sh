# dataset-fed LeNet step, A/B over throw-away switches:  gpurun -- 'bash tools/experiments/feed_ab.sh "T4_X_WRAP=0 T4_X_HITDEV=0" "T4_X_WRAP=1 T4_X_HITDEV=0" ...'
cd /tmp && rm -rf fp && mkdir fp && cd fp && python3 $GRAFT_REPO_ROOT/tools/make_synth_mnist.py data/MNIST/raw 8192 256 > /dev/null
S=$GRAFT_REPO_ROOT/tools/forth/lenet_dataset_epoch_nohit.4th
sed 's/^3 epochs/12 epochs/; s/ms_for_3_epochs/ms_for_12_epochs/' $S > /tmp/fp/e12.4th
for rep in 1 2 3; do
  for cfg in "$@"; do
    echo "$cfg: $(env $cfg T4_SEED=1 timeout 120 $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < /tmp/fp/e12.4th | grep -o 'ms_for_12_epochs [0-9.]* hits [0-9]*')"
  done
done
