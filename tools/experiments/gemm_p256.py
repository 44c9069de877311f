#!/usr/bin/env python3
"""k_gemm_plain256 (256 x 256 tiles, 16 waves) against k_gemm_plain128: exactness on integer operands + launch time, T4K_GEMM_PLAIN256 = 0 | 1 read once per process."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(); k.init(0)
p = lambda t: t.data_ptr()
def timeit(fn, iters):
    e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p()
    k.call("t4k_event_create", ctypes.byref(e0)); k.call("t4k_event_create", ctypes.byref(e1))
    for _ in range(iters): fn()
    best = 1e9
    for _ in range(4):
        k.call("t4k_event_record", e0, None)
        for _ in range(iters): fn()
        k.call("t4k_event_record", e1, None); k.call("t4k_event_sync", e1)
        ms = ctypes.c_float(0); k.call("t4k_event_elapsed_ms", e0, e1, ctypes.byref(ms))
        best = min(best, ms.value / iters * 1e3)
    return best
print("T4K_GEMM_PLAIN256 =", os.environ.get("T4K_GEMM_PLAIN256", "1 (default)"))
g = torch.Generator(device="cuda"); g.manual_seed(3)
for M, N, K, tA, tB in ((4096, 4096, 1024, 0, 0), (4096, 4096, 1024, 0, 1), (4096, 4096, 1024, 1, 0), (4096, 4096, 1024, 1, 1), (4096, 4096, 4096, 0, 0), (8192, 4096, 512, 0, 0), (8192, 8192, 1024, 0, 1)):
    A = torch.randint(-2, 3, (K, M) if tA else (M, K), device="cuda", generator=g).float(); B = torch.randint(-2, 3, (N, K) if tB else (K, N), device="cuda", generator=g).float()
    O = torch.zeros(M, N, device="cuda"); torch.cuda.synchronize()
    k.call("t4k_gemm", p(A), p(B), p(O), 1.0, 0.0, tA, tB, M, N, K, 1, None); k.call("t4k_sync", None)
    ref = (A.t() if tA else A).double() @ (B.t() if tB else B).double()
    exact = bool(torch.equal(O.double(), ref))
    t = timeit(lambda: k.call("t4k_gemm", p(A), p(B), p(O), 1.0, 0.0, tA, tB, M, N, K, 1, None), max(20, int(0.15 / (2.0 * M * N * K / 100e12))))
    print("%5d %5d %5d tA=%d tB=%d  %8.2f us  %6.1f TFLOP/s  %5.1f %%  exact=%s" % (M, N, K, tA, tB, t, 2.0 * M * N * K / t / 1e6, 2.0 * M * N * K / t / 1e6 / 157.3 * 100, exact), flush=True)
