#!/usr/bin/env python3
"""t4k_gemm time against K for a fixed output (fixed cost vs per-stage cost of the kernel the dispatcher picks):  gemm_k_sweep.py M N tA tB K..."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(); k.init(0)
p = lambda t: t.data_ptr()
def timeit(fn, iters=100):
    e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p()
    k.call("t4k_event_create", ctypes.byref(e0)); k.call("t4k_event_create", ctypes.byref(e1))
    for _ in range(10): fn()
    k.call("t4k_event_record", e0, None)
    for _ in range(iters): fn()
    k.call("t4k_event_record", e1, None); k.call("t4k_event_sync", e1)
    ms = ctypes.c_float(0); k.call("t4k_event_elapsed_ms", e0, e1, ctypes.byref(ms))
    return ms.value / iters * 1e3
M, N, tA, tB = (int(x) for x in sys.argv[1:5])
for K in (int(x) for x in sys.argv[5:]):
    A = torch.rand(M * K, device="cuda"); B = torch.rand(K * N, device="cuda"); O = torch.zeros(M, N, device="cuda")
    t = timeit(lambda: k.call("t4k_gemm", p(A), p(B), p(O), 1.0, 0.0, tA, tB, M, N, K, 1, None))
    print("M=%d N=%d K=%5d tA=%d tB=%d: %7.2f us  %6.1f TFLOP/s" % (M, N, K, tA, tB, t, 2.0 * M * N * K / t / 1e6), flush=True)
