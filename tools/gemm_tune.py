#!/usr/bin/env python3
"""Time t4k_gemm variants (T4K_GEMM_VARIANT) on 1024^3 with HIP events; check against torch."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def one(shape=(1024, 1024, 1024), iters=200):
    import torch
    from tensorforth_amd.lib import load
    k = load(); k.init(0); k.call("t4k_set_default_stream", None)
    M, N, K = shape
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    A = torch.rand(M, K, device="cuda", generator=g) - 0.5; B = torch.rand(K, N, device="cuda", generator=g) - 0.5
    O = torch.zeros(M, N, device="cuda")
    args = (A.data_ptr(), B.data_ptr(), O.data_ptr(), 1.0, 0.0, 0, 0, M, N, K, 1, None)
    k.call("t4k_gemm", *args); torch.cuda.synchronize()
    err = float((O - A @ B).abs().max() / (A @ B).abs().max())
    e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p()
    k.call("t4k_event_create", ctypes.byref(e0)); k.call("t4k_event_create", ctypes.byref(e1))
    for _ in range(20): k.call("t4k_gemm", *args)
    best = 1e9; reps = []
    for _ in range(8):
        k.call("t4k_event_record", e0, None)
        for _ in range(iters): k.call("t4k_gemm", *args)
        k.call("t4k_event_record", e1, None); k.call("t4k_event_sync", e1)
        ms = ctypes.c_float(); k.call("t4k_event_elapsed_ms", e0, e1, ctypes.byref(ms))
        best = min(best, ms.value / iters); reps.append(round(ms.value / iters * 1e3, 2))
    tf = 2.0 * M * N * K / (best * 1e-3) / 1e12
    print("variant=%s shape=%s  %.2f us  %.1f TFLOP/s (%.1f%% of 157.3)  relerr=%.2e" %
          (os.environ.get("T4K_GEMM_VARIANT", "default"), shape, best * 1e3, tf, 100 * tf / 157.3, err), "reps_us", reps, flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        shp = tuple(int(x) for x in sys.argv[2].split("x")) if len(sys.argv) > 2 else (1024, 1024, 1024)
        one(shp)
    else:
        for v in (sys.argv[1:] or ["0", "1", "2", "3"]):
            env = dict(os.environ, T4K_GEMM_VARIANT=v)
            subprocess.call([sys.executable, __file__, "one"], env=env)
        subprocess.call([sys.executable, __file__, "one", "2048x2048x2048"])
        subprocess.call([sys.executable, __file__, "one", "4096x4096x4096"])
