#!/usr/bin/env python3
"""Write synthetic MNIST-shaped IDX files where the reference's loader expects them (src/ld/mnist.cpp:21-62:
magic 0x00000803 / 0x00000801, big-endian counts): class-dependent Gaussian bump + uint8 noise, seed 42,
so a training run's loss visibly falls.  usage: make_synth_mnist.py [root=./data/MNIST/raw] [n_train=8192] [n_test=1024]"""
import os, struct, sys
import numpy as np

root = sys.argv[1] if len(sys.argv) > 1 else "./data/MNIST/raw"
n_train = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
n_test = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
os.makedirs(root, exist_ok=True)
rng = np.random.default_rng(42)
yy, xx = np.mgrid[0:28, 0:28]


def make(n, prefix):
    lab = rng.integers(0, 10, n).astype(np.uint8)
    cx = 6 + 2 * (lab % 5)[:, None, None] * 1.0 + 0 * xx; cy = 8 + 6 * (lab // 5)[:, None, None] * 1.0 + 0 * yy
    img = 200.0 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / 18.0) + rng.integers(0, 40, (n, 28, 28))
    img = np.clip(img, 0, 255).astype(np.uint8)
    with open(os.path.join(root, prefix + "-images-idx3-ubyte"), "wb") as f:
        f.write(struct.pack(">IIII", 0x00000803, n, 28, 28)); f.write(img.tobytes())
    with open(os.path.join(root, prefix + "-labels-idx1-ubyte"), "wb") as f:
        f.write(struct.pack(">II", 0x00000801, n)); f.write(lab.tobytes())


make(n_train, "train"); make(n_test, "t10k")
print("wrote %d train / %d test images under %s" % (n_train, n_test, root))
