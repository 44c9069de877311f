"""do two INDEPENDENT training processes on one GPU disturb each other?  (each a world-1 job: no exchange between them)"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vm_util import rel_err
from lenet_parity import PARAMS
W = os.path.join(ROOT, "tests", "xchg_worker.py")
def run(n, envx={}):
    ds = [tempfile.mkdtemp() for _ in range(n)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **envx)
    ps = [subprocess.Popen([sys.executable, W, d, "0", "1", "32", "6"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for d in ds]
    outs = [p.communicate(timeout=200)[0] for p in ps]
    assert not any(p.returncode for p in ps), outs
    return [np.load(os.path.join(d, "out0.npz")) for d in ds]
solo = run(1)[0]
for trial in range(4):
    for envx in ({}, {"T4_HEAD_BWD": "0"}, {"T4_STACK_HEAD": "0"}, {"T4_STACK": "0"}):
        res = run(3, envx)
        print(trial, envx, " | ".join(" ".join("%s=%.0e" % (n_, rel_err(r[n_], solo[n_])) for n_, _ in PARAMS[:8:2]) for r in res), flush=True)
