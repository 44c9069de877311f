#!/usr/bin/env python3
"""Per-stage and fixed cost of the dF partial kernel: 8x8 images, 128 -> 256 channels (18 tiles of 128 x 128, 14 pixel slices), batch 128 ... 1024 =
   10 ... 74 stages of 64 pixels per workgroup; the launch time of k_convbig_dfw alone comes from rocprofv3 (conv_df_pmc.sh), here: the whole dF|dB call."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(); k.init(0)
p = lambda t: t.data_ptr()
def timeit(fn, iters=200):
    e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p()
    k.call("t4k_event_create", ctypes.byref(e0)); k.call("t4k_event_create", ctypes.byref(e1))
    for _ in range(iters): fn()
    best = 1e9
    for _ in range(3):
        k.call("t4k_event_record", e0, None)
        for _ in range(iters): fn()
        k.call("t4k_event_record", e1, None); k.call("t4k_event_sync", e1)
        ms = ctypes.c_float(0); k.call("t4k_event_elapsed_ms", e0, e1, ctypes.byref(ms))
        best = min(best, ms.value / iters * 1e3)
    return best
H, C1, C0 = 8, 128, 256
S, T = [], []
for N in (128, 256, 384, 512, 768, 1024):
    x = torch.rand(N, H, H, C1, device="cuda") - 0.5; f = torch.rand(C1, 3, 3, C0, device="cuda") - 0.5; y = torch.rand(N, H, H, C0, device="cuda") - 0.5
    df = torch.zeros_like(f); db = torch.zeros(C0, device="cuda"); torch.cuda.synchronize()
    t = timeit(lambda: k.call("t4k_conv2d_bwd", p(x), p(y), None, p(f), p(df), p(db), N, H, H, C1, H, H, C0, 3, 1, 1, 1, None))
    st = -(-(N * H * H // 14) // 64)
    S.append(st); T.append(t)
    print("N=%4d ~%2d stages: dF|dB %.1f us (%.0f%% of the MFMA peak)" % (N, st, t, 100 * 2.0 * N * H * H * C1 * C0 * 9 / t / 1e6 / 157.3), flush=True)
a, b = np.polyfit(np.array(S, float), np.array(T), 1)
print("%.3f us per stage (MFMA time 3.413) + %.2f us fixed (incl. fold + column sums)" % (a, b))
