#!/usr/bin/env python3
"""The CIFAR net's many-channel convolutions against the dense product of the same size (M = pixels, N = Cout, K = 9 Cin), same data
   distribution (uniform random: the clock a launch gets depends on the operand bits), same box, back to back."""
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
for (N, H, C1, C0) in [(256, 16, 64, 128), (256, 8, 128, 256)]:
    for dist in ("rand", "zeros"):
        mk = (lambda *s: torch.rand(*s, device="cuda") - 0.5) if dist == "rand" else (lambda *s: torch.zeros(*s, device="cuda"))
        x = mk(N, H, H, C1); f = mk(C1, 3, 3, C0); b = mk(C0); y = mk(N, H, H, C0); dx = torch.zeros_like(x); df = torch.zeros_like(f); db = torch.zeros_like(b)
        M = N * H * H; K = 9 * C1
        A = mk(M, K); B = mk(K, C0); O = torch.zeros(M, C0, device="cuda")
        A2 = mk(M, 9 * C0); B2 = mk(9 * C0, C1); O2 = torch.zeros(M, C1, device="cuda")
        torch.cuda.synchronize()
        flop = 2.0 * M * K * C0
        tf = timeit(lambda: k.call("t4k_conv2d_fwd", p(x), p(y), p(f), p(b), N, H, H, C1, H, H, C0, 3, 1, 1, None))
        tdx = timeit(lambda: k.call("t4k_conv2d_bwd", p(x), p(y), p(dx), p(f), None, None, N, H, H, C1, H, H, C0, 3, 1, 1, 0, None))
        tdf = timeit(lambda: k.call("t4k_conv2d_bwd", p(x), p(y), None, p(f), p(df), p(db), N, H, H, C1, H, H, C0, 3, 1, 1, 1, None))
        tg = timeit(lambda: k.call("t4k_gemm", p(A), p(B), p(O), 1.0, 0.0, 0, 0, M, C0, K, 1, None))
        tg2 = timeit(lambda: k.call("t4k_gemm", p(A2), p(B2), p(O2), 1.0, 0.0, 0, 0, M, C1, 9 * C0, 1, None))
        pc = lambda t: 100 * flop / t / 1e6 / 157.3
        print("%-5s N=%d %dx%d %d->%d: fwd %.1f us (%.0f%%)  dX %.1f (%.0f%%)  dF|dB %.1f (%.0f%%) | gemm %dx%dx%d %.1f us (%.0f%%)  gemm %dx%dx%d %.1f (%.0f%%)" %
              (dist, N, H, H, C1, C0, tf, pc(tf), tdx, pc(tdx), tdf, pc(tdf), M, C0, K, tg, pc(tg), M, C1, 9 * C0, tg2, pc(tg2)), flush=True)
