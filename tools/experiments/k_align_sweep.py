#!/usr/bin/env python3
"""Does the 3136-byte row pitch of the 784-wide GAN layers cost anything by itself?  Same products with K / N = 768, 784, 800, 832 (rows of
3072 / 3136 / 3200 / 3328 bytes: 784 and 800 put every other row 64 bytes off the 128-byte lines).  N = 256 rows."""
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
N = 256
for wide in (704, 768, 784, 800, 832, 896):
    # discriminator's first layer: E1 = wide -> 512;  generator's last: 512 -> wide
    for E1, E0 in ((wide, 512), (512, wide)):
        X = torch.rand(N, E1, device="cuda") - 0.5; W = torch.rand(E0, E1, device="cuda") - 0.5; b = torch.rand(E0, device="cuda"); Y = torch.zeros(N, E0, device="cuda")
        t = timeit(lambda: k.call("t4k_linear_fwd", p(X), p(W), p(b), p(Y), N, E0, E1, None))
        dY = torch.rand(N, E0, device="cuda") - 0.5; dX = torch.zeros(N, E1, device="cuda"); dW = torch.zeros(E0, E1, device="cuda"); dB = torch.zeros(E0, device="cuda")
        t2 = timeit(lambda: k.call("t4k_gemm", p(dY), p(W), p(dX), 1.0, 0.0, 0, 0, N, E1, E0, 1, None))
        t3 = timeit(lambda: k.call("t4k_gemm", p(dY), p(X), p(dW), 1.0, 0.0, 1, 0, E0, E1, N, 1, None))
        t4 = timeit(lambda: k.call("t4k_linear_bwd", p(X), p(W), p(dY), p(dX), p(dW), p(dB), N, E0, E1, 1, None))
        print("E1=%4d E0=%4d  fwd %6.2f  dX %6.2f  dW %6.2f  dual(dW||dX) %6.2f us   MFLOP %d" % (E1, E0, t, t2, t3, t4, 2 * N * E1 * E0 // 1000000), flush=True)
