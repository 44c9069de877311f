#!/usr/bin/env python3
"""t4k_gemm / t4k_linear_fwd launch time over the four operand layouts and an alpha / beta / bias epilogue:  gemm_layouts.py M N K [K ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(); k.init(0)
p = lambda t: t.data_ptr()
def timeit(fn, iters=200):
    e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p()
    k.call("t4k_event_create", ctypes.byref(e0)); k.call("t4k_event_create", ctypes.byref(e1))
    for _ in range(20): fn()
    best = 1e9
    for _ in range(5):
        k.call("t4k_event_record", e0, None)
        for _ in range(iters): fn()
        k.call("t4k_event_record", e1, None); k.call("t4k_event_sync", e1)
        ms = ctypes.c_float(0); k.call("t4k_event_elapsed_ms", e0, e1, ctypes.byref(ms))
        best = min(best, ms.value / iters * 1e3)
    return best
M, N = int(sys.argv[1]), int(sys.argv[2])
for K in (int(x) for x in sys.argv[3:]):
    A = torch.rand(M * K, device="cuda") - 0.5; B = torch.rand(K * N, device="cuda") - 0.5; O = torch.zeros(M, N, device="cuda"); bias = torch.rand(N, device="cuda")
    row = []
    for tA, tB in ((0, 0), (0, 1), (1, 0), (1, 1)):
        row.append("t%d%d %6.2f" % (tA, tB, timeit(lambda: k.call("t4k_gemm", p(A), p(B), p(O), 1.0, 0.0, tA, tB, M, N, K, 1, None))))
    row.append("t01 a2 b-1 %6.2f" % timeit(lambda: k.call("t4k_gemm", p(A), p(B), p(O), 2.0, -1.0, 0, 1, M, N, K, 1, None)))
    row.append("linear+bias %6.2f" % timeit(lambda: k.call("t4k_linear_fwd", p(A), p(B), p(bias), p(O), M, N, K, None)))
    print("M=%d N=%d K=%5d us: %s   (%.1f TFLOP/s at t00)" % (M, N, K, "  ".join(row), 2.0 * M * N * K / float(row[0].split()[1]) / 1e6), flush=True)
