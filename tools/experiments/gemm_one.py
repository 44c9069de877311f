#!/usr/bin/env python3
"""One GEMM shape, ~0.6 s of launches - the command rocprofv3 wraps for the per-shape averages in profiles/:
   gemm_one.py M N K tA tB [alpha beta] [bias]   (bias: t4k_linear_fwd, i.e. tA=0 tB=1 + bias)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(); k.init(0)
p = lambda t: t.data_ptr()
M, N, K, tA, tB = (int(x) for x in sys.argv[1:6])
al, be = (float(sys.argv[6]), float(sys.argv[7])) if len(sys.argv) > 7 else (1.0, 0.0)
bias = "bias" in sys.argv
A = torch.rand(M * K, device="cuda") - 0.5; B = torch.rand(K * N, device="cuda") - 0.5; O = torch.zeros(M, N, device="cuda"); b = torch.rand(N, device="cuda")
# the part's clocks ramp up over the first few hundred ms of sustained MFMA load: ~0.6 s of launches, the summary averages the LAST 500
n_launch = max(520, int(0.6 / (2.0 * M * N * K / 100e12)))
for _ in range(n_launch):
    if bias: k.call("t4k_linear_fwd", p(A), p(B), p(b), p(O), M, N, K, None)
    else:    k.call("t4k_gemm", p(A), p(B), p(O), al, be, tA, tB, M, N, K, 1, None)
torch.cuda.synchronize()
