"""bisect the multi-process exchange: per-tensor error of the ranks' result against ONE VM on the whole batch, under env variants"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vm_util import rel_err
from lenet_parity import PARAMS, _get, _setup
from tensorforth_amd.vm import VM
world, rows, steps = int(sys.argv[1]), 32, int(sys.argv[2]) if len(sys.argv) > 2 else 3
whole = VM(device=0, seed=505); _setup(whole, world * rows, 0, world * rows)
for _ in range(steps):
    whole.eval("net fw bw opt drop\n")
ref = {n_: _get(whole, e) for n_, e in PARAMS}
for envx in ({}, {"T4_OPT_FOLD": "0"}, {"T4_DP_XCHG": "0"}, {"T4_LAZY_DX0": "0"}, {"T4K_STACK_SPLIT": "1", "T4K_STACK_BSPLIT": "1"}, {}):
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **envx)
        ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "xchg_worker.py"), d, str(r), str(world), str(rows), str(steps)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        outs = [p.communicate(timeout=200)[0] for p in ps]
        if any(p.returncode for p in ps):
            print(envx, "FAILED", [o[-300:] for o in outs]); continue
        res = [np.load(os.path.join(d, "out%d.npz" % r)) for r in range(world)]
        same = all(np.array_equal(res[0][n_], res[r][n_]) for r in range(1, world) for n_, _ in PARAMS)
        print(envx, "replicas_identical", same, "launches", float(res[0]["launches"]), " ".join("%s=%.1e" % (n_, rel_err(res[0][n_], ref[n_])) for n_, _ in PARAMS), flush=True)
