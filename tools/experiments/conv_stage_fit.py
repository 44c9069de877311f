#!/usr/bin/env python3
"""Per-stage and fixed cost of k_convbig8 against k_gemm_plain128 on the SAME product: a 1x1 convolution over 16384 pixels, Cin = 64 s channels, 256 out
   (256 tiles of 128 x 128, one per CU) for s = 2 ... 36 stages; least-squares line through (s, time)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(); k.init(0)
p = lambda t: t.data_ptr()
def timeit(fn, iters=300):
    e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p()
    k.call("t4k_event_create", ctypes.byref(e0)); k.call("t4k_event_create", ctypes.byref(e1))
    for _ in range(iters): fn()
    best = 1e9
    for _ in range(3):
        k.call("t4k_event_record", e0, None)
        for _ in range(iters): fn()
        k.call("t4k_event_record", e1, None); k.call("t4k_event_sync", e1)
        ms = ctypes.c_float(0); k.call("t4k_event_elapsed_ms", e0, e1, ctypes.byref(ms))
        best = min(best, ms.value / iters * 1e3)
    return best
N, H, C0 = int(os.environ.get("NB", "256")), 8, int(os.environ.get("C0", "256"))
S = (2, 4, 8, 12, 18, 24, 36)
tc, tg = [], []
for s in S:
    C1 = 64 * s; M = N * H * H
    x = torch.rand(N, H, H, C1, device="cuda") - 0.5; f = torch.rand(C1, 1, 1, C0, device="cuda") - 0.5; b = torch.zeros(C0, device="cuda"); y = torch.zeros(N, H, H, C0, device="cuda")
    torch.cuda.synchronize()
    tc.append(timeit(lambda: k.call("t4k_conv2d_fwd", p(x), p(y), p(f), p(b), N, H, H, C1, H, H, C0, 1, 1, 0, None)))
    tg.append(timeit(lambda: k.call("t4k_gemm", p(x), p(f), p(y), 1.0, 0.0, 0, 0, M, C0, C1, 1, None)))
    print("stages %2d: conv %.2f us  gemm %.2f us" % (s, tc[-1], tg[-1]), flush=True)
for name, t in (("conv", tc), ("gemm", tg)):
    a, b0 = np.polyfit(np.array(S, float), np.array(t), 1)
    print("%s: %.3f us per stage (MFMA time of a stage at 2.4 GHz: 3.413 us) + %.2f us fixed" % (name, a, b0))
