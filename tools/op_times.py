#!/usr/bin/env python3
"""Per-op launch cost of the LeNet (L128) training step: each C-ABI op is issued `iters` times back to
back on one stream between two HIP events => (kernel time + in-order dispatch boundary) per call."""
import ctypes
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensorforth_amd import lib as t4lib

k = t4lib.load(); k.init(0)
N = 128
z = lambda *s: torch.rand(*s, device="cuda")
p = lambda t: t.data_ptr()


class PoolBlock(ctypes.Structure):
    _fields_ = [("pre_layer", ctypes.c_int), ("pre_alpha", ctypes.c_float), ("pre_mask", ctypes.c_void_p), ("pre_out", ctypes.c_void_p),
                ("pool_layer", ctypes.c_int), ("KS", ctypes.c_int), ("pool_out", ctypes.c_void_p),
                ("post_layer", ctypes.c_int), ("post_alpha", ctypes.c_float), ("post_mask", ctypes.c_void_p), ("post_out", ctypes.c_void_p),
                ("copy_out", ctypes.c_void_p)]


def timeit(name, fn, iters=300):
    e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p()
    k.call("t4k_event_create", ctypes.byref(e0)); k.call("t4k_event_create", ctypes.byref(e1))
    for _ in range(10):
        fn()
    k.call("t4k_event_record", e0, None)
    for _ in range(iters):
        fn()
    k.call("t4k_event_record", e1, None); k.call("t4k_event_sync", e1)
    ms = ctypes.c_float(0); k.call("t4k_event_elapsed_ms", e0, e1, ctypes.byref(ms))
    print("%-34s %7.2f us" % (name, ms.value / iters * 1e3), flush=True)
    return ms.value / iters * 1e3


x0 = z(N, 28, 28, 1); f1 = z(1, 3, 3, 10) - 0.5; b1 = z(10); c1 = z(N, 28, 28, 10)
p1 = z(N, 14, 14, 10); r1 = z(N, 14, 14, 10); m1 = z(N, 14, 14, 10)
f2 = z(10, 3, 3, 20) - 0.5; b2 = z(20); c2 = z(N, 14, 14, 20); d2 = z(N, 14, 14, 20); dm2 = z(N, 14, 14, 20)
p2 = z(N, 7, 7, 20); r2 = z(N, 7, 7, 20); m2 = z(N, 7, 7, 20); fl = z(N, 980)
w3 = z(100, 980) - 0.5; b3 = z(100); y3 = z(N, 100); w4 = z(10, 100) - 0.5; b4 = z(10); y4 = z(N, 10); sm = z(N, 10)
df1 = torch.zeros_like(f1); db1 = torch.zeros_like(b1); df2 = torch.zeros_like(f2); db2 = torch.zeros_like(b2)
dw3 = torch.zeros_like(w3); db3 = torch.zeros_like(b3); dw4 = torch.zeros_like(w4); db4 = torch.zeros_like(b4)
dx0 = z(N, 28, 28, 1); dx1 = z(N, 14, 14, 10); gx3 = z(N, 980); gx4 = z(N, 100)
tot = 0.0
T = lambda n, f: timeit(n, f)
tot += T("copy 100K (n0=input)", lambda: k.call("t4k_copy", p(x0), p(dx0), x0.numel(), None))
tot += T("conv1 fwd 1->10", lambda: k.call("t4k_conv2d_fwd", p(x0), p(c1), p(f1), p(b1), N, 28, 28, 1, 28, 28, 10, 3, 1, 1, None))
blk1 = PoolBlock(); blk1.KS = 2; blk1.pool_layer = 14; blk1.pool_out = p(p1); blk1.post_layer = 4; blk1.post_mask = p(m1); blk1.post_out = p(r1)
tot += T("run1 fwd pool+relu", lambda: k.call("t4k_poolblock_fwd", p(c1), ctypes.byref(blk1), N, 28, 28, 14, 14, 10, None))
tot += T("conv2 fwd 10->20", lambda: k.call("t4k_conv2d_fwd", p(r1), p(c2), p(f2), p(b2), N, 14, 14, 10, 14, 14, 20, 3, 1, 1, None))
blk2 = PoolBlock(); blk2.KS = 2; blk2.pre_layer = 10; blk2.pre_alpha = 0.5; blk2.pre_mask = p(dm2); blk2.pre_out = p(d2)
blk2.pool_layer = 14; blk2.pool_out = p(p2); blk2.post_layer = 4; blk2.post_mask = p(m2); blk2.post_out = p(r2); blk2.copy_out = p(fl)
tot += T("run2 fwd drop+pool+relu+flat", lambda: k.call("t4k_poolblock_fwd", p(c2), ctypes.byref(blk2), N, 14, 14, 7, 7, 20, None))
tot += T("linear1 fwd 980->100", lambda: k.call("t4k_linear_fwd", p(fl), p(w3), p(b3), p(y3), N, 100, 980, None))
tot += T("rand 12800", lambda: k.call("t4k_rand", p(gx4), 12800, 0, 0.0, 1.0, None))
tot += T("activate dropout 12800", lambda: k.call("t4k_activate", 10, p(y3), p(gx4), p(gx4), 0.5, 12800, None))
tot += T("linear2 fwd 100->10", lambda: k.call("t4k_linear_fwd", p(y3), p(w4), p(b4), p(y4), N, 10, 100, None))
tot += T("softmax 128x10", lambda: k.call("t4k_softmax", p(y4), p(sm), N, 10, None))
print("forward total %.1f us" % tot); fw = tot
tot += T("tt_op sub 1280", lambda: k.call("t4k_tt_op", 17, p(sm), p(y4), p(sm), 1280, None))
tot += T("copy 1280", lambda: k.call("t4k_copy", p(sm), p(y4), 1280, None))
tot += T("linear2 bwd (dB,dW,dX)", lambda: k.call("t4k_linear_bwd", p(y3), p(w4), p(y4), p(gx4), p(dw4), p(db4), N, 10, 100, 1, None))
tot += T("  linear2 dX only", lambda: k.call("t4k_linear_bwd", p(y3), p(w4), p(y4), p(gx4), None, None, N, 10, 100, 0, None)) * 0
tot += T("tt_op mul 12800", lambda: k.call("t4k_tt_op", 18, p(gx4), p(y3), p(gx4), 12800, None))
tot += T("linear1 bwd (dB,dW,dX)", lambda: k.call("t4k_linear_bwd", p(fl), p(w3), p(y3), p(gx3), p(dw3), p(db3), N, 100, 980, 1, None))
tot += T("  linear1 dX only", lambda: k.call("t4k_linear_bwd", p(fl), p(w3), p(y3), p(gx3), None, None, N, 100, 980, 0, None)) * 0
tot += T("run2 bwd", lambda: k.call("t4k_poolblock_bwd", p(gx3), p(c2), ctypes.byref(blk2), N, 14, 14, 7, 7, 20, None))
T("  conv2 bwd dF|dB only", lambda: k.call("t4k_conv2d_bwd", p(r1), p(c2), None, p(f2), p(df2), p(db2), N, 14, 14, 10, 14, 14, 20, 3, 1, 1, 1, None))
T("  conv2 bwd dX only", lambda: k.call("t4k_conv2d_bwd", p(r1), p(c2), p(dx1), p(f2), None, None, N, 14, 14, 10, 14, 14, 20, 3, 1, 1, 0, None))
tot += T("conv2 bwd2 (dF|dB, dX x2)", lambda: k.call("t4k_conv2d_bwd2", p(r1), p(c2), p(dx1), p(r1), p(f2), p(df2), p(db2), N, 14, 14, 10, 14, 14, 20, 3, 1, 1, 1, None))
tot += T("run1 bwd", lambda: k.call("t4k_poolblock_bwd", p(dx1), p(c1), ctypes.byref(blk1), N, 28, 28, 14, 14, 10, None))
T("  conv1 bwd dF|dB only", lambda: k.call("t4k_conv2d_bwd", p(x0), p(c1), None, p(f1), p(df1), p(db1), N, 28, 28, 1, 28, 28, 10, 3, 1, 1, 1, None))
T("  conv1 bwd dX only", lambda: k.call("t4k_conv2d_bwd", p(x0), p(c1), p(dx0), p(f1), None, None, N, 28, 28, 1, 28, 28, 10, 3, 1, 1, 0, None))
tot += T("conv1 bwd2 (dF|dB, dX x2)", lambda: k.call("t4k_conv2d_bwd2", p(x0), p(c1), p(dx0), p(x0), p(f1), p(df1), p(db1), N, 28, 28, 1, 28, 28, 10, 3, 1, 1, 1, None))
tot += T("sgd 98000 (stand-in for opt)", lambda: k.call("t4k_sgd", p(w3), p(dw3), None, 1, 0.01, 0.0, 98000, None))
print("backward+opt total %.1f us, step total %.1f us" % (tot - fw, tot))
timeit("empty-ish: memset 4B", lambda: k.call("t4k_memset", p(db1), 0, 4, None))
