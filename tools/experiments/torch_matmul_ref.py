#!/usr/bin/env python3
"""The vendor library on the same box, for the DESIGN.md comparison line: torch.mm / torch.addmm (rocBLAS / hipBLASLt behind PyTorch-ROCm, fp32,
no TF32-style down-conversion) timed exactly like tools/experiments/gemm_layouts.py times t4k_gemm.  Not part of the product or of any test."""
import sys
import torch
torch.backends.cuda.matmul.allow_tf32 = False
def timeit(fn, iters=200):
    for _ in range(20): fn()
    best = 1e9
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
print("torch", torch.__version__, torch.cuda.get_device_name(0), "preferred blas:", getattr(torch.backends.cuda, "preferred_blas_library", lambda: "?")())
for M, N, K in [(1024, 1024, 1024), (1024, 1024, 784), (512, 1024, 1024), (2048, 2048, 2048), (2048, 2048, 784), (4096, 4096, 1024), (4096, 4096, 4096), (16384, 256, 1152), (65536, 128, 576)]:
    A = torch.rand(M, K, device="cuda") - 0.5; B = torch.rand(K, N, device="cuda") - 0.5; W = torch.rand(N, K, device="cuda") - 0.5; b = torch.rand(N, device="cuda"); O = torch.empty(M, N, device="cuda")
    t_nn = timeit(lambda: torch.mm(A, B, out=O))
    t_nt = timeit(lambda: torch.mm(A, W.t(), out=O))
    t_lin = timeit(lambda: torch.addmm(b, A, W.t(), out=O))
    f = 2.0 * M * N * K / 1e6
    print("M=%d N=%d K=%d  mm NN %7.2f us (%5.1f TFLOP/s)   mm NT %7.2f us (%5.1f)   addmm bias NT %7.2f us (%5.1f)" % (M, N, K, t_nn, f / t_nn, t_nt, f / t_nt, t_lin, f / t_lin), flush=True)
