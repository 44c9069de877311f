"""diagnostic: where do the product VM and the oracle VM part ways in the config-#4 GAN round (full fp32 tensors)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from vm_util import OracleVM, rel_err
import test_gpu_baseline_configs as T
from tensorforth_amd.vm import VM
g, o = VM(device=0, seed=31), OracleVM(seed=31)
src = T._body("cfg4_gan256", "D 2 rounds")
for vm in (g, o): vm.eval(src)
def cmp(tag, m, e):
    a, b = T._fetch(g, m, e), T._fetch(o, m, e)
    d = np.abs(a.astype(np.float64) - b); s = np.abs(b).max()
    print("%-28s %-10s max|ref| %.3e  max|d| %.3e  rel %.2e  frac>1e-4 %.4f" % (tag, m + " " + e, s, d.max(), d.max() / max(s, 1e-30), (d > 1e-4 * s).mean()), flush=True)
for vm in (g, o): vm.eval("D 1 trainable real forward REAL backprop\n")
cmp("D after real bp", "D", "0 nn.dw"); cmp("", "D", "6 nn.dw")
for vm in (g, o): vm.eval("F\n")
for vm in (g, o): vm.eval("forward\n")
cmp("D fwd(F)", "D", "-1 n@"); cmp("G out", "G", "-1 n@"); cmp("G h1", "G", "2 n@")
for vm in (g, o): vm.eval("FAKE backprop\n")
cmp("D after fake bp", "D", "0 nn.dw")
for vm in (g, o): vm.eval("0.0001 0.5 nn.adam\n")
for e in ("0 nn.w", "3 nn.w", "6 nn.w", "6 nn.b"): cmp("D after adam", "D", e)
for vm in (g, o): vm.eval("0 trainable F forward REAL backprop\n")
cmp("D frozen bp dx", "D", "0 n@"); cmp("", "D", "3 n@"); cmp("", "D", "6 n@"); cmp("", "D", "7 n@")
for vm in (g, o): vm.eval("0 n@ G swap backprop\n")
for e in ("4 nn.dw", "4 nn.db", "2 nn.dw", "0 nn.dw", "0 nn.db", "4 n@", "2 n@"): cmp("G bp", "G", e)
a, b = T._fetch(g, "G", "0 nn.dw"), T._fetch(o, "G", "0 nn.dw")
print("G dw0 |ref| quantiles", np.quantile(np.abs(b), [0.01, 0.1, 0.5, 0.9, 0.99, 1.0]))
for vm in (g, o): vm.eval("0.0004 0.5 nn.adam drop\n")
for e in ("0 nn.w", "2 nn.w", "4 nn.w", "4 nn.b"): cmp("G after adam", "G", e)
print("---- round 2", flush=True)
for vm in (g, o): vm.eval("D 1 trainable real forward REAL backprop F forward FAKE backprop\n")
for e in ("0 nn.dw", "3 nn.dw", "6 nn.dw", "6 nn.db"): cmp("D r2 grads", "D", e)
for vm in (g, o): vm.eval("0.0001 0.5 nn.adam\n")
for e in ("0 nn.w", "3 nn.w", "6 nn.w", "6 nn.b"): cmp("D r2 after adam", "D", e)
for vm in (g, o): vm.eval("0 trainable F\n")
pre_g, pre_o = T._fetch(g, "G", "3 n@"), T._fetch(o, "G", "3 n@")      # input of the second leakyrelu = linear-2 output
print("G pre-activation L3: min |x| oracle", np.abs(pre_o).min(), "gpu", np.abs(pre_g).min(), "sign flips", int((np.sign(pre_g) != np.sign(pre_o)).sum()), flush=True)
for vm in (g, o): vm.eval("forward REAL backprop 0 n@ G swap backprop\n")
for e in ("4 nn.dw", "4 nn.db", "2 nn.dw", "0 nn.dw", "0 nn.db"): cmp("G r2 bp", "G", e)
for e in ("1 nn.ex", "3 nn.ex"):
    a, b = T._fetch(g, "G", e), T._fetch(o, "G", e)
    bad = np.argwhere(a != b); print("G", e, "mask mismatches:", len(bad), bad[:4].tolist(), [(a[tuple(i)], b[tuple(i)]) for i in bad[:4]], flush=True)
for vm in (g, o): vm.eval("0.0004 0.5 nn.adam drop\n")
for e in ("0 nn.w", "0 nn.b", "2 nn.w", "4 nn.w", "4 nn.b"): cmp("G r2 after adam", "G", e)
a, b = T._fetch(g, "G", "0 nn.w"), T._fetch(o, "G", "0 nn.w")
d = (a.astype(np.float64) - b).ravel(); print("G w0 diff quantiles", np.quantile(np.abs(d), [0.5, 0.9, 0.99, 1.0]))
