#!/usr/bin/env python3
"""conv2d fwd / bwd timing at CIFAR-class sizes (many channels => MFMA implicit-GEMM path) against the fp32 MFMA peak."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(os.environ.get("T4K_LIB")); k.init(0)
p = lambda t: t.data_ptr()


def timeit(fn, iters=60):
    e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p()
    k.call("t4k_event_create", ctypes.byref(e0)); k.call("t4k_event_create", ctypes.byref(e1))
    for _ in range(10): fn()                    # first touches of freshly allocated 67 MB tensors are slow (3 were not enough)
    k.call("t4k_event_record", e0, None)
    for _ in range(iters): fn()
    k.call("t4k_event_record", e1, None); k.call("t4k_event_sync", e1)
    ms = ctypes.c_float(0); k.call("t4k_event_elapsed_ms", e0, e1, ctypes.byref(ms))
    return ms.value / iters * 1e3


# clocks ramp up over the first few hundred ms of sustained MFMA load: warm the part before timing anything
_x = torch.rand(256, 32, 32, 64, device="cuda"); _f = torch.rand(64, 3, 3, 64, device="cuda"); _b = torch.rand(64, device="cuda"); _y = torch.zeros(256, 32, 32, 64, device="cuda")
timeit(lambda: k.call("t4k_conv2d_fwd", p(_x), p(_y), p(_f), p(_b), 256, 32, 32, 64, 32, 32, 64, 3, 1, 1, None), iters=1500)
for (N, H, C1, C0) in [(256, 32, 64, 64), (256, 16, 64, 128), (256, 32, 3, 64), (128, 14, 10, 20)]:
    x = torch.rand(N, H, H, C1, device="cuda"); f = torch.rand(C1, 3, 3, C0, device="cuda") - 0.5; b = torch.rand(C0, device="cuda")
    y = torch.zeros(N, H, H, C0, device="cuda"); dx = torch.zeros_like(x); df = torch.zeros_like(f); db = torch.zeros_like(b)
    flop = 2.0 * N * H * H * C1 * C0 * 9
    tf = timeit(lambda: k.call("t4k_conv2d_fwd", p(x), p(y), p(f), p(b), N, H, H, C1, H, H, C0, 3, 1, 1, None))
    tb = timeit(lambda: k.call("t4k_conv2d_bwd", p(x), p(y), p(dx), p(f), p(df), p(db), N, H, H, C1, H, H, C0, 3, 1, 1, 1, None))
    tdf = timeit(lambda: k.call("t4k_conv2d_bwd", p(x), p(y), None, p(f), p(df), p(db), N, H, H, C1, H, H, C0, 3, 1, 1, 1, None))
    tdx = timeit(lambda: k.call("t4k_conv2d_bwd", p(x), p(y), p(dx), p(f), None, None, N, H, H, C1, H, H, C0, 3, 1, 1, 0, None))
    print("N=%d %dx%d %d->%d: fwd %.1f us (%.1f TF, %.0f%% of peak)   bwd (dF+dX) %.1f us (%.1f TF)   dF|dB %.1f us  dX %.1f us" %
          (N, H, H, C1, C0, tf, flop / tf / 1e6, 100 * flop / tf / 1e6 / 157.3, tb, 2 * flop / tb / 1e6, tdf, tdx), flush=True)
