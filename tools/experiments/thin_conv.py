#!/usr/bin/env python3
"""Image-input convolution 3 -> 64 at the CIFAR net's size (256 x 32 x 32), forward and dF | dB: us per launch (wall clock, back to back)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(); k.init(0)
p = lambda t: t.data_ptr()
N, H, C1, C0 = 256, 32, 3, 64
X = torch.randn(N, H, H, C1, device="cuda"); F = torch.randn(C1, 3, 3, C0, device="cuda"); B = torch.randn(C0, device="cuda")
Y = torch.zeros(N, H, H, C0, device="cuda"); XC = torch.zeros_like(X); G = torch.randn_like(Y); DF = torch.zeros_like(F); DB = torch.zeros_like(B)
def t(fn, n=300):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("fwd  %.2f us" % t(lambda: k.call("t4k_conv2d_fwd2", p(X), p(XC), p(Y), p(F), p(B), N, H, H, C1, H, H, C0, 3, 1, 1, None)))
print("dF   %.2f us" % t(lambda: k.call("t4k_conv2d_bwd", p(X), p(G), None, p(F), p(DF), p(DB), N, H, H, C1, H, H, C0, 3, 1, 1, 1, None)))
print("fwd (no layer-0 copy)  %.2f us" % t(lambda: k.call("t4k_conv2d_fwd2", p(X), None, p(Y), p(F), p(B), N, H, H, C1, H, H, C0, 3, 1, 1, None)))
X4 = torch.randn(N, H, H, 4, device="cuda"); F4 = torch.randn(4, 3, 3, C0, device="cuda")
print("fwd 4 -> 64  %.2f us" % t(lambda: k.call("t4k_conv2d_fwd2", p(X4), None, p(Y), p(F4), p(B), N, H, H, 4, H, H, C0, 3, 1, 1, None)))
X1 = torch.randn(N, H, H, 1, device="cuda"); F1 = torch.randn(1, 3, 3, C0, device="cuda")
print("fwd 1 -> 64  %.2f us" % t(lambda: k.call("t4k_conv2d_fwd2", p(X1), None, p(Y), p(F1), p(B), N, H, H, 1, H, H, C0, 3, 1, 1, None)))
Y.zero_(); torch.cuda.synchronize()
print("memset 67 MB  %.2f us" % t(lambda: Y.zero_()))
print("copy 67 MB  %.2f us" % t(lambda: Y.copy_(G)))
