import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
import vm_util
vm_util.ELEM_MIN = 0.0
import numpy as np
import lenet_parity as lp
_orig = vm_util.check_tensor
LOG = {}
def chk(name, got, want, tol=1e-4, elem_min=0.0):
    _orig(name, got, want, tol, 0.0)
    for fl in (1e-3, 1e-2):
        f = vm_util.elem_frac(got, want, rtol=tol, floor=fl)
        key = (name.split(" ", 2)[-1] if name.startswith("step") else name, fl)
        LOG[key] = min(f, LOG.get(key, 1.0))
    LOG[(name.split(" ", 2)[-1] if name.startswith("step") else name, "n")] = np.asarray(want).size
lp.check_tensor = chk
from tensorforth_amd.vm import VM
from vm_util import OracleVM
g, o = VM(device=0, seed=2024), OracleVM(seed=2024)
for vm in (g, o): lp._setup(vm, 128, 0, 128)
img = g.fetch("img"); g.eval("drop")
for step in range(3): lp.step_vs_oracle(g, o, img, step)
names = sorted({k[0] for k in LOG})
for n in names: print("%-70s n=%-8d floor1e-3: %.5f   floor1e-2: %.5f" % (n, LOG[(n, "n")], LOG[(n, 1e-3)], LOG[(n, 1e-2)]))
