"""flakiness census of the multi-process exchange: the standard worker flow, several runs per env variant"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vm_util import rel_err
from lenet_parity import PARAMS, _get, _setup
from tensorforth_amd.vm import VM
world, rows, steps, reps = 2, 32, int(os.environ.get("DBG_STEPS", "3")), int(sys.argv[1]) if len(sys.argv) > 1 else 5
whole = VM(device=0, seed=505); _setup(whole, world * rows, 0, world * rows)
for _ in range(steps):
    whole.eval("net fw bw opt drop\n")
ref = {n_: _get(whole, e) for n_, e in PARAMS}
whole.close()
variants = [{}]
rows = int(os.environ.get("DBG_ROWS", "32"))
for envx in variants:
    errs = []
    for rep in range(reps):
        with tempfile.TemporaryDirectory() as d:
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **envx)
            ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "xchg_worker.py"), d, str(r), str(world), str(rows), str(steps)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
            outs = [p.communicate(timeout=200)[0] for p in ps]
            if any(p.returncode for p in ps):
                errs.append("FAIL:" + " || ".join(o[-160:].replace("\n", " ") for o in outs)); continue
            res = [np.load(os.path.join(d, "out%d.npz" % r)) for r in range(world)]
            same = all(np.array_equal(res[0][n_], res[r][n_]) for r in range(1, world) for n_, _ in PARAMS)
            errs.append("%.0e%s" % (max(rel_err(res[0][n_], ref[n_]) for n_, _ in PARAMS), "" if same else "(replicas differ)"))
    print(envx, errs, flush=True)
