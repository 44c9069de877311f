#!/usr/bin/env python3
"""Times t4k_conv2d_block_fwd (LeNet conv2: 128x14x14x10 -> 20, dropout + 2x2 maxpool + relu + flatten) with the library
given in T4K_LIB (see conv_ablate_build.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(os.environ.get("T4K_LIB")); k.init(0)
N = 128
z = lambda *s: torch.rand(*s, device="cuda")
p = lambda t: t.data_ptr()
class PoolBlock(ctypes.Structure):
    _fields_ = [("pre_layer", ctypes.c_int), ("pre_alpha", ctypes.c_float), ("pre_mask", ctypes.c_void_p), ("pre_out", ctypes.c_void_p),
                ("pool_layer", ctypes.c_int), ("KS", ctypes.c_int), ("pool_out", ctypes.c_void_p),
                ("post_layer", ctypes.c_int), ("post_alpha", ctypes.c_float), ("post_mask", ctypes.c_void_p), ("post_out", ctypes.c_void_p),
                ("copy_out", ctypes.c_void_p)]
def timeit(fn, iters=400):
    e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p()
    k.call("t4k_event_create", ctypes.byref(e0)); k.call("t4k_event_create", ctypes.byref(e1))
    for _ in range(50): fn()
    k.call("t4k_event_record", e0, None)
    for _ in range(iters): fn()
    k.call("t4k_event_record", e1, None); k.call("t4k_event_sync", e1)
    ms = ctypes.c_float(0); k.call("t4k_event_elapsed_ms", e0, e1, ctypes.byref(ms))
    return ms.value / iters * 1e3
r1 = z(N, 14, 14, 10); f2 = z(10, 3, 3, 20) - 0.5; b2 = z(20); c2 = z(N, 14, 14, 20); d2 = z(N, 14, 14, 20); dm2 = z(N, 14, 14, 20)
p2 = z(N, 7, 7, 20); r2 = z(N, 7, 7, 20); m2 = z(N, 7, 7, 20); fl = z(N, 980)
blk2 = PoolBlock(); blk2.KS = 2; blk2.pre_layer = 10; blk2.pre_alpha = 0.5; blk2.pre_mask = p(dm2); blk2.pre_out = p(d2)
blk2.pool_layer = 14; blk2.pool_out = p(p2); blk2.post_layer = 4; blk2.post_mask = p(m2); blk2.post_out = p(r2); blk2.copy_out = p(fl)
t = timeit(lambda: k.call("t4k_conv2d_block_fwd", p(r1), None, p(c2), p(f2), p(b2), ctypes.byref(blk2), N, 14, 14, 10, 14, 14, 20, 3, 1, 1, None))
t2 = timeit(lambda: k.call("t4k_conv2d_fwd", p(r1), p(c2), p(f2), p(b2), N, 14, 14, 10, 14, 14, 20, 3, 1, 1, None))
dx1 = z(N, 14, 14, 10); df2 = torch.zeros_like(f2); db2 = torch.zeros_like(b2)
t3 = timeit(lambda: k.call("t4k_conv2d_bwd", p(r1), p(c2), p(dx1), p(f2), p(df2), p(db2), N, 14, 14, 10, 14, 14, 20, 3, 1, 1, 1, None))
print("%-10s conv2 block fwd %6.2f us   plain fwd %6.2f us   bwd(dF+dX+fold) %6.2f us" % (os.path.basename(os.environ.get("T4K_LIB", "product")), t, t2, t3), flush=True)
# conv1 block (image-input layer 1 -> 10 channels, 2x2 maxpool + relu behind it, layer-0 copy)
x0 = z(N, 28, 28, 1); xc = z(N, 28, 28, 1); f1 = z(1, 3, 3, 10) - 0.5; b1 = z(10); c1 = z(N, 28, 28, 10)
p1 = z(N, 14, 14, 10); q1 = z(N, 14, 14, 10); m1 = z(N, 14, 14, 10)
blk1 = PoolBlock(); blk1.KS = 2; blk1.pool_layer = 14; blk1.pool_out = p(p1); blk1.post_layer = 4; blk1.post_mask = p(m1); blk1.post_out = p(q1)
t4 = timeit(lambda: k.call("t4k_conv2d_block_fwd", p(x0), p(xc), p(c1), p(f1), p(b1), ctypes.byref(blk1), N, 28, 28, 1, 28, 28, 10, 3, 1, 1, None))
print("conv1 block fwd (copy + conv + pool + relu) %6.2f us  [T4K_CONV_FEW=%s]" % (t4, os.environ.get("T4K_CONV_FEW", "1")), flush=True)
