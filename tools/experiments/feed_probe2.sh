cd /tmp && rm -rf fp && mkdir fp && cd fp && python3 $GRAFT_REPO_ROOT/tools/make_synth_mnist.py data/MNIST/raw 8192 256 > /dev/null
for i in 1 2 3; do T4_SEED=1 $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $GRAFT_REPO_ROOT/tools/forth/lenet_dataset_epoch_nohit.4th | grep -o "ms_for_3_epochs [0-9.]*"; done
