#!/usr/bin/env python3
"""Time t4k_mlp_head_bwd (k_head_bwd_l32) alone, with the LAB build's ablation bits (T4K_HB_LAB, wrong results - timing only):
   T4K_LIB=tensorforth_amd/libt4hip_lab.so T4K_HB_LAB=1 python tools/experiments/head_bwd_lab.py
   bits: 1 no column riders, 2 no tiles, 4 no target-store counter, 8 tiles skip P/T/W2 loads, 64 tiles skip mask loads, 16 tiles stop behind the prologue, 32 no in-place gate"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(os.environ.get("T4K_LIB") or None); k.init(0)
k.call("t4k_set_default_stream", None)
p = lambda t: t.data_ptr()
N, E1, EA, EB = [int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (128, 980, 100, 10))]
g = lambda *s: torch.rand(*s, device="cuda") - 0.5
X1, W1, X2, W2, P, T, M = g(N, E1), g(EA, E1), g(N, EA), g(EB, EA), g(N, EB), g(N, EB), (torch.rand(N, EA, device="cuda") > 0.5).float()
Y1, Y2, DW1, DB1, DW2, DB2 = g(N, EA), g(N, EB), g(EA, E1), g(EA), g(EB, EA), g(EB)
def once():
    k.call("t4k_mlp_head_bwd", p(X2), p(W2), p(P), p(T), p(Y2), p(M), p(Y1), p(DW2), p(DB2), p(X1), p(W1), p(DW1), p(DB1), N, E1, EA, EB, None)
best = 1e9
for rep in range(4):
    for _ in range(300): once()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000): once()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 2000 * 1e6)
print("T4K_HB_LAB=%s  N=%d E1=%d EA=%d EB=%d: %.2f us per launch (back to back)" % (os.environ.get("T4K_HB_LAB", "0"), N, E1, EA, EB, best), flush=True)
