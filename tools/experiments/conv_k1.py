#!/usr/bin/env python3
"""k_convbig8 as a plain GEMM: a 1x1 convolution with 9 Cin channels does the same arithmetic as the 3x3 layer with Cin (same stages, no taps outside the image)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(); k.init(0)
p = lambda t: t.data_ptr()
def timeit(fn, iters=200):
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
for (N, H, C1, C0, K) in [(256, 8, 128, 256, 3), (256, 8, 1152, 256, 1), (512, 8, 128, 256, 3), (256, 16, 64, 128, 3), (256, 16, 576, 128, 1), (256, 16, 128, 64, 3), (256, 16, 1152, 64, 1), (256, 8, 256, 128, 3), (256, 8, 2304, 128, 1)]:
    x = torch.rand(N, H, H, C1, device="cuda") - 0.5; f = torch.rand(C1, K, K, C0, device="cuda") - 0.5; b = torch.rand(C0, device="cuda"); y = torch.zeros(N, H, H, C0, device="cuda")
    torch.cuda.synchronize()
    flop = 2.0 * N * H * H * C1 * C0 * K * K
    tf = timeit(lambda: k.call("t4k_conv2d_fwd", p(x), p(y), p(f), p(b), N, H, H, C1, H, H, C0, K, 1, K // 2, None))
    print("N=%d %dx%d %d->%d K=%d: fwd %.1f us (%.0f%%)" % (N, H, H, C1, C0, K, tf, 100 * flop / tf / 1e6 / 157.3), flush=True)
