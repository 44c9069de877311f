# dataset-fed LeNet loop with the prefetch ring on / off, then its per-kernel table:  gpurun -- 'bash tools/experiments/feed_probe.sh'
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/tools/make_synth_mnist.py /tmp/data/MNIST/raw 8192 1024 > /dev/null
S=$GRAFT_REPO_ROOT/tools/forth/lenet_dataset_epoch_nohit.4th
for p in 1 0 1 0; do echo "T4_FEED_PREFETCH=$p"; T4_FEED_PREFETCH=$p $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $S | grep -i "ms_\|per"; done
rm -rf /tmp/kd; rocprofv3 --kernel-trace --stats -d /tmp/kd -o ds -- $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $S > /tmp/kd.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/kd -name "*.db" | head -1) | head -14 | cut -c1-70,112-150
