#!/usr/bin/env python3
"""Write synthetic CIFAR-10-shaped binary batches where the reference's loader expects them (src/ld/loader.cpp:41-46,
src/ld/cifar10.cpp:90-135: records of 1 label byte + 3 x 32 x 32 PLANAR colour bytes, R plane then G then B):
./data/CIFAR10/cifar-10-batches-bin/{data_batch.bin, test_batch.bin}.  Each class is a coloured Gaussian blob at a class-dependent
place over uint8 noise (seed 7), so a training run's loss visibly falls; pixel (y, x, c) of sample i is reproducible from the seed,
which lets a test check the loader's planar -> HWC re-ordering value by value.
usage: make_synth_cifar.py [root=./data/CIFAR10/cifar-10-batches-bin] [n_train=2048] [n_test=512]"""
import os
import sys

import numpy as np


def synth(n, seed):
    """labels [n] uint8 and images [n, 3, 32, 32] uint8 (planar, as stored in the file)"""
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, 10, n).astype(np.uint8)
    yy, xx = np.mgrid[0:32, 0:32]
    cx = (6 + 5 * (lab % 5))[:, None, None]; cy = (9 + 14 * (lab // 5))[:, None, None]
    blob = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / 30.0)                       # [n,32,32]
    tint = np.stack([(lab * 37) % 200 + 55, (lab * 91) % 200 + 55, (lab * 53) % 200 + 55], 1).astype(np.float64)   # [n,3]
    img = blob[:, None, :, :] * tint[:, :, None, None] + rng.integers(0, 48, (n, 3, 32, 32))
    return lab, np.clip(img, 0, 255).astype(np.uint8)


def write(path, lab, img):
    rec = np.concatenate([lab[:, None], img.reshape(len(lab), -1)], axis=1).astype(np.uint8)   # 1 + 3072 bytes per sample
    with open(path, "wb") as f:
        f.write(rec.tobytes())


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "./data/CIFAR10/cifar-10-batches-bin"
    n_train = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    n_test = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    os.makedirs(root, exist_ok=True)
    write(os.path.join(root, "data_batch.bin"), *synth(n_train, 7))
    write(os.path.join(root, "test_batch.bin"), *synth(n_test, 8))
    print("wrote %d train / %d test CIFAR-10-shaped records under %s" % (n_train, n_test, root))
