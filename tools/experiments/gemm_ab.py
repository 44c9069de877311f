#!/usr/bin/env python3
"""A/B of two builds of the library on a list of GEMM shapes (wall clock over back-to-back launches after a clock ramp):
   T4K_LIB=tensorforth_amd/libt4hip_alt.so python tools/experiments/gemm_ab.py            (default: the release library)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(os.environ.get("T4K_LIB") or None); k.init(0)
p = lambda t: t.data_ptr()
SHAPES = [(1024, 1024, 1024, 0, 0), (1024, 1024, 1024, 0, 1), (1024, 1024, 1024, 1, 0), (1024, 1024, 1024, 1, 1), (1024, 1024, 784, 0, 1),
          (2048, 2048, 1024, 0, 1), (2048, 2048, 2048, 0, 0), (2048, 2048, 784, 0, 1), (4096, 4096, 1024, 0, 1), (512, 1024, 1024, 0, 1)]
for M, N, K, tA, tB in SHAPES:
    A = torch.rand(M * K, device="cuda") - 0.5; B = torch.rand(K * N, device="cuda") - 0.5; O = torch.zeros(M, N, device="cuda")
    n = max(520, int(0.5 / (2.0 * M * N * K / 100e12)))
    best = 1e9
    for rep in range(3):
        for _ in range(n // 2): k.call("t4k_gemm", p(A), p(B), p(O), 1.0, 0.0, tA, tB, M, N, K, 1, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): k.call("t4k_gemm", p(A), p(B), p(O), 1.0, 0.0, tA, tB, M, N, K, 1, None)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n * 1e6)
    print("%5d %5d %5d tA=%d tB=%d  %8.2f us  %6.1f TFLOP/s  %5.1f %%" % (M, N, K, tA, tB, best, 2.0 * M * N * K / best / 1e6, 2.0 * M * N * K / best / 1e6 / 157.3 * 100), flush=True)
