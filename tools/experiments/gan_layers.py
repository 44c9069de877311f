#!/usr/bin/env python3
"""Launch time + max error of the GAN's layer products (N = 256 rows) through t4k_linear_fwd / t4k_gemm:  gan_layers.py
   (env switches are read once per process: run it once per setting, e.g. T4K_GEMM_L32=0 / 1)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(); k.init(0)
p = lambda t: t.data_ptr()
def timeit(fn, iters=100):
    e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p()
    k.call("t4k_event_create", ctypes.byref(e0)); k.call("t4k_event_create", ctypes.byref(e1))
    for _ in range(10): fn()
    best = 1e9
    for _ in range(4):
        k.call("t4k_event_record", e0, None)
        for _ in range(iters): fn()
        k.call("t4k_event_record", e1, None); k.call("t4k_event_sync", e1)
        ms = ctypes.c_float(0); k.call("t4k_event_elapsed_ms", e0, e1, ctypes.byref(ms))
        best = min(best, ms.value / iters * 1e3)
    return best
N = 256
torch.manual_seed(1)
print("switches:", {e: os.environ[e] for e in os.environ if e.startswith("T4K_") or e.startswith("T4_")})
for E1, E0 in ((784, 512), (512, 256), (128, 256), (256, 512), (512, 784), (784, 100), (100, 784), (1000, 300)):
    X = torch.rand(N, E1, device="cuda") - 0.5; W = torch.rand(E0, E1, device="cuda") - 0.5; b = torch.rand(E0, device="cuda"); Y = torch.zeros(N, E0, device="cuda")
    t = timeit(lambda: k.call("t4k_linear_fwd", p(X), p(W), p(b), p(Y), N, E0, E1, None))
    ref = (X.double().cpu() @ W.double().cpu().T + b.double().cpu()); err = ((Y.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    # backward products: dX[N,E1] = dY[N,E0] W[E0,E1] (t00), dW[E0,E1] = dY^T[E0,N] X[N,E1] (t10)
    dY = torch.rand(N, E0, device="cuda") - 0.5; dX = torch.zeros(N, E1, device="cuda"); dW = torch.zeros(E0, E1, device="cuda")
    t2 = timeit(lambda: k.call("t4k_gemm", p(dY), p(W), p(dX), 1.0, 0.0, 0, 0, N, E1, E0, 1, None))
    e2 = ((dX.double().cpu() - dY.double().cpu() @ W.double().cpu()).abs().max()).item()
    t3 = timeit(lambda: k.call("t4k_gemm", p(dY), p(X), p(dW), 1.0, 0.0, 1, 0, E0, E1, N, 1, None))
    e3 = ((dW.double().cpu() - dY.double().cpu().T @ X.double().cpu()).abs().max()).item()
    print("E1=%4d E0=%4d  fwd %6.2f us (rel err %.1e)   dX %6.2f us (abs err %.1e)   dW %6.2f us (abs err %.1e)" % (E1, E0, t, err, t2, e2, t3, e3), flush=True)
