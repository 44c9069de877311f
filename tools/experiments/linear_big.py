#!/usr/bin/env python3
"""Linear forward / backward at MLP sizes: TFLOP/s of t4k_linear_fwd and the in-place t4k_linear_bwd (dW, dB, dX)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(); k.init(0)
p = lambda t: t.data_ptr()
def timeit(fn, iters=50):
    e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p()
    k.call("t4k_event_create", ctypes.byref(e0)); k.call("t4k_event_create", ctypes.byref(e1))
    for _ in range(5): fn()
    k.call("t4k_event_record", e0, None)
    for _ in range(iters): fn()
    k.call("t4k_event_record", e1, None); k.call("t4k_event_sync", e1)
    ms = ctypes.c_float(0); k.call("t4k_event_elapsed_ms", e0, e1, ctypes.byref(ms))
    return ms.value / iters * 1e3
shapes = [(512, 1024, 1024), (1024, 4096, 4096), (2048, 1024, 4096), (4096, 4096, 1024), (256, 2048, 2048), (1024, 1000, 4096)]
if len(sys.argv) > 1: shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for (N, E0, E1) in shapes:
    X = torch.rand(N, E1, device="cuda"); W = torch.rand(E0, E1, device="cuda") * 1e-2; B = torch.zeros(E0, device="cuda")
    Y = torch.zeros(N, E0, device="cuda"); G = torch.rand(N, E0, device="cuda") * 1e-3
    DW = torch.zeros(E0, E1, device="cuda"); DB = torch.zeros(E0, device="cuda")
    tf = timeit(lambda: k.call("t4k_linear_fwd", p(X), p(W), p(B), p(Y), N, E0, E1, None))
    tb = timeit(lambda: k.call("t4k_linear_bwd", p(X), p(W), p(G), p(X), p(DW), p(DB), N, E0, E1, 1, None))
    fl = 2.0 * N * E0 * E1
    print("N=%5d %4d<-%4d: fwd %8.1f us %6.1f TFLOP/s (%4.1f %%)   bwd %8.1f us %6.1f TFLOP/s (%4.1f %%)" %
          (N, E0, E1, tf, fl / tf / 1e6, fl / tf / 1e6 / 1.573, tb, 2 * fl / tb / 1e6, 2 * fl / tb / 1e6 / 1.573), flush=True)
