#!/usr/bin/env python3
"""One conv backward (dF only) shape, repeated - the command rocprofv3 wraps for per-kernel time / PMC of the many-channel dF kernel:
   conv_df_one.py N H C1 C0 [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(); k.init(0)
p = lambda t: t.data_ptr()
N, H, C1, C0 = (int(x) for x in sys.argv[1:5]); iters = int(sys.argv[5]) if len(sys.argv) > 5 else 300
x = torch.rand(N, H, H, C1, device="cuda"); f = torch.rand(C1, 3, 3, C0, device="cuda") - 0.5
y = torch.rand(N, H, H, C0, device="cuda"); df = torch.zeros_like(f); db = torch.zeros(C0, device="cuda")
for _ in range(iters):
    k.call("t4k_conv2d_bwd", p(x), p(y), None, p(f), p(df), p(db), N, H, H, C1, H, H, C0, 3, 1, 1, 1, None)
torch.cuda.synchronize()
