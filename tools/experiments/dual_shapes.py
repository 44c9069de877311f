#!/usr/bin/env python3
"""Back-to-back time of the linear backward (dW += dY^T X, dB, dX in place) for the GAN nets' layers: the dual launch against the separate kernels."""
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
shapes = [(256, 256, 512), (256, 512, 784), (256, 784, 512), (256, 512, 256), (256, 256, 128)]
if len(sys.argv) > 1: shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for (N, E0, E1) in shapes:
    X = torch.rand(N, E1, device="cuda"); W = torch.rand(E0, E1, device="cuda"); G = torch.rand(N, E0, device="cuda") * 1e-3
    DW = torch.zeros(E0, E1, device="cuda"); DB = torch.zeros(E0, device="cuda")
    tb = timeit(lambda: k.call("t4k_linear_bwd", p(X), p(W), p(G), p(X), p(DW), p(DB), N, E0, E1, 1, None))
    DX = torch.zeros(N, E1, device="cuda")
    tn = timeit(lambda: k.call("t4k_linear_bwd", p(X), p(W), p(G), p(DX), p(DW), p(DB), N, E0, E1, 1, None))
    fl = 2.0 * N * E0 * E1 * 2
    print("N=%4d %3d<-%3d: bwd in place %6.2f us  (%5.1f TFLOP/s)   dX to its own buffer %6.2f us" % (N, E0, E1, tb, fl / tb / 1e6, tn), flush=True)
