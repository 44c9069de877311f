#!/usr/bin/env python3
"""CPU baseline worker (TEST / MEASUREMENT INFRASTRUCTURE ONLY - never on the product path).

bench.py's `cpu_baseline` leg starts one copy of this script per host core it wants to load (plain
subprocesses: the bench process holds a HIP runtime and must not fork) and adds up what they report:

    cpu_baseline.py step <net> <batch> <seconds>   -> prints "<training steps done> <elapsed seconds>"
        the oracle's restatement of `forward backprop 0.01 nn.sgd` on one synthetic batch, repeated
    cpu_baseline.py gemm <rows> <n> <k> <seconds>  -> prints "<slab products done> <elapsed seconds>"
        the reference's blocked host GEMM (src/mu/tensor.cu:97-123, restated in t4_oracle.cpp) on a
        [rows, k] x [k, n] slab, repeated; C workers with rows = 1024 / C make up 1024^3 products
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    import numpy as np
    import t4oracle
    from tensorforth_amd import pymodel          # layer lists only (nn_f / nn_c builders are backend-agnostic)
    mode = sys.argv[1]
    if mode == "step":
        net, n, secs = sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
        build = pymodel.nn_c if net == "nn_c" else pymodel.nn_f
        rng = np.random.default_rng(42 + os.getpid() % 1000)
        lab = rng.integers(0, 10, n).astype(np.uint32)
        x = rng.random((n, 28, 28, 1)).astype(np.float32)
        om = build(t4oracle.OracleModel(n, 28, 28, 1, seed=1234))
        om.forward(x); om.onehot_labels(lab)
        t0 = time.perf_counter(); nst = 0
        while True:
            om.forward(x); om.backprop(); om.sgd(0.01, 0.0); nst += 1
            if time.perf_counter() - t0 > secs:
                break
        print(nst, time.perf_counter() - t0)
    elif mode == "gemm":
        rows, n, k, secs = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
        rng = np.random.default_rng(1)
        a = rng.random((rows, k)).astype(np.float32); b = rng.random((k, n)).astype(np.float32)
        o = np.zeros((rows, n), np.float32)
        t0 = time.perf_counter(); cnt = 0
        while True:
            t4oracle.lib().t4o_gemm_host_blocked(t4oracle.P(a), t4oracle.P(b), t4oracle.P(o), 1.0, 0.0, rows, n, k); cnt += 1
            if time.perf_counter() - t0 > secs:
                break
        print(cnt, time.perf_counter() - t0)
    else:
        raise SystemExit("usage: cpu_baseline.py step|gemm ...")


if __name__ == "__main__":
    main()
