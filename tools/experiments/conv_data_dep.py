#!/usr/bin/env python3
"""Is the 64->64 forward (225 us) vs dX (187 us) gap of the same-shape conv_big kernel data dependent?  Times both with
the operand tensors swapped / zeroed."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(); k.init(0)
p = lambda t: t.data_ptr()
def timeit(fn, iters=60):
    e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p()
    k.call("t4k_event_create", ctypes.byref(e0)); k.call("t4k_event_create", ctypes.byref(e1))
    for _ in range(5): fn()
    k.call("t4k_event_record", e0, None)
    for _ in range(iters): fn()
    k.call("t4k_event_record", e1, None); k.call("t4k_event_sync", e1)
    ms = ctypes.c_float(0); k.call("t4k_event_elapsed_ms", e0, e1, ctypes.byref(ms))
    return ms.value / iters * 1e3
N, H, C = 256, 32, 64
rand = torch.rand(N, H, H, C, device="cuda"); normal = torch.randn(N, H, H, C, device="cuda") * 3; zero = torch.zeros(N, H, H, C, device="cuda")
f = torch.rand(C, 3, 3, C, device="cuda") - 0.5; b = torch.rand(C, device="cuda"); out = torch.zeros(N, H, H, C, device="cuda")
timeit(lambda: k.call("t4k_conv2d_fwd", p(rand), p(out), p(f), p(b), N, H, H, C, H, H, C, 3, 1, 1, None), iters=1000)
for name, a in (("rand(0,1)", rand), ("normal*3", normal), ("zeros", zero)):
    tf = timeit(lambda: k.call("t4k_conv2d_fwd", p(a), p(out), p(f), p(b), N, H, H, C, H, H, C, 3, 1, 1, None))
    tx = timeit(lambda: k.call("t4k_conv2d_bwd", p(rand), p(a), p(out), p(f), None, None, N, H, H, C, H, H, C, 3, 1, 1, 0, None))
    print("gathered operand %-10s fwd %.1f us   dX %.1f us" % (name, tf, tx), flush=True)
