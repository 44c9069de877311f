#!/usr/bin/env python3
"""Stand-alone time of the head-sized linear backward (k_linsmall_bwd) for a few shapes, incl. the GAN discriminator's 256 -> 1 layer."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(); k.init(0)
p = lambda t: t.data_ptr()
def timeit(fn, iters=300):
    e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p()
    k.call("t4k_event_create", ctypes.byref(e0)); k.call("t4k_event_create", ctypes.byref(e1))
    for _ in range(20): fn()
    k.call("t4k_event_record", e0, None)
    for _ in range(iters): fn()
    k.call("t4k_event_record", e1, None); k.call("t4k_event_sync", e1)
    ms = ctypes.c_float(0); k.call("t4k_event_elapsed_ms", e0, e1, ctypes.byref(ms))
    return ms.value / iters * 1e3
for (N, E0, E1) in [(128, 10, 100), (256, 1, 256), (256, 10, 256), (256, 1, 128), (1024, 10, 100), (256, 64, 256)]:
    X = torch.rand(N, E1, device="cuda"); W = torch.rand(E0, E1, device="cuda"); G = torch.rand(N, E0, device="cuda")
    DW = torch.zeros(E0, E1, device="cuda"); DB = torch.zeros(E0, device="cuda"); Y = torch.zeros(N, E0, device="cuda"); B = torch.zeros(E0, device="cuda")
    tb = timeit(lambda: k.call("t4k_linear_bwd", p(X), p(W), p(G), p(X), p(DW), p(DB), N, E0, E1, 1, None))
    tf = timeit(lambda: k.call("t4k_linear_fwd", p(X), p(W), p(B), p(Y), N, E0, E1, None))
    print("N=%4d %3d<-%3d: bwd (dB,dW,dX in place) %6.2f us   fwd %6.2f us" % (N, E0, E1, tb, tf), flush=True)
