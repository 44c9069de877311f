import sys, os, ctypes, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from tensorforth_amd import lib as t4lib
k = t4lib.load(); k.init(0); k.call("t4k_set_default_stream", None)
o = ctypes.CDLL(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle", "libt4oracle.so"))
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
p = lambda t: t.data_ptr()
rng = np.random.default_rng(5); n = 5000
for kind in ("sgd0", "sgdm", "adam", "adamw"):
    W = rng.standard_normal(n).astype(np.float32); G = rng.standard_normal(n).astype(np.float32)
    M = (rng.standard_normal(n) * 0.1).astype(np.float32); V = (rng.random(n) * 0.1).astype(np.float32)
    dW, dG, dM, dV = (torch.from_numpy(x.copy()).cuda() for x in (W, G, M, V))
    F = ctypes.c_float
    if kind == "sgd0": o.t4o_sgd(P(W), P(G), P(M), 3, F(0.01), F(0.0), ctypes.c_long(n)); k.call("t4k_sgd", p(dW), p(dG), p(dM), 3, 0.01, 0.0, n, None)
    elif kind == "sgdm": o.t4o_sgd(P(W), P(G), P(M), 1, F(0.01), F(0.9), ctypes.c_long(n)); k.call("t4k_sgd", p(dW), p(dG), p(dM), 1, 0.01, 0.9, n, None)
    elif kind == "adam": o.t4o_adam(P(W), P(G), P(M), P(V), F(1e-3), F(0.9), F(0.999), ctypes.c_long(n)); k.call("t4k_adam", p(dW), p(dG), p(dM), p(dV), 1e-3, 0.9, 0.999, n, None)
    else: o.t4o_adamw(P(W), P(G), P(M), P(V), F(1e-3), F(0.9), F(0.999), F(0.01), ctypes.c_long(n)); k.call("t4k_adamw", p(dW), p(dG), p(dM), p(dV), 1e-3, 0.9, 0.999, 0.01, n, None)
    k.call("t4k_sync", None)
    for nm, a, b in (("W", dW.cpu().numpy(), W), ("M", dM.cpu().numpy(), M), ("V", dV.cpu().numpy(), V)):
        d = a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)
        print(kind, nm, "mismatches", int((d != 0).sum()), "max ulp", int(np.abs(d).max()))
