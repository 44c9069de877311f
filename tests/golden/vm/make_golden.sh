#!/bin/sh
# Regenerates tests/golden/vm/*.out: the output of the oracle-backed VM (oracle/ten4_oracle, CPU)
# on the Forth scripts under tests/scripts/ with T4_SEED=1.  test_vm_scripts.py checks (on CPU) that
# ten4_oracle still reproduces these files and (on the GPU) that the HIP-backed `ten4` matches them.
cd "$(dirname "$0")/../../.." || exit 1
make -C oracle ten4_oracle >/dev/null || exit 1
ROOT=$(pwd)
WORK=$(mktemp -d)                       # the dataset words read ./data/MNIST/raw relative to the working directory
python3 tools/make_synth_mnist.py "$WORK/data/MNIST/raw" 1024 256 >/dev/null || exit 1
python3 tools/make_synth_cifar.py "$WORK/data/CIFAR10/cifar-10-batches-bin" 256 64 >/dev/null || exit 1
for s in tests/scripts/*.4th; do
    n=$(basename "$s" .4th)
    (cd "$WORK" && T4_SEED=1 "$ROOT/oracle/ten4_oracle" < "$ROOT/$s" > "$ROOT/tests/golden/vm/$n.out") || exit 1
done
rm -rf "$WORK"
