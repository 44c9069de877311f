"""diagnostic: accuracy of the product's conv dF / dB against float64 on the product's OWN operands, by batch size"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from vm_util import rel_err
import test_gpu_config5_full as C
from tensorforth_amd.vm import VM
for N in (128, 256, 512, 1024):
    vm = VM(device=0, seed=505)
    C._setup(vm, N, 0, N)
    img = vm.fetch("img"); vm.eval("drop")
    vm.eval("net fw\n"); x3 = C._get(vm, "3 n@"); vm.eval("bw\n")
    do0, do3 = C._get(vm, "1 n@"), C._get(vm, "4 n@")
    for tag, X, dO, L in (("conv1", img, do0, 0), ("conv2", x3, do3, 3)):
        dF, dB = C.conv_df64(X, dO)
        gF, gB = C._get(vm, "%d nn.dw" % L), C._get(vm, "%d nn.db" % L)
        d = np.abs(gF.astype(np.float64) - dF.reshape(gF.shape))
        print("N=%4d %s dF rel %.2e (max|ref| %.3g, worst at %s)  dB rel %.2e" % (N, tag, rel_err(gF, dF.reshape(gF.shape)), np.abs(dF).max(), np.unravel_index(d.argmax(), d.shape), rel_err(gB.ravel(), dB)), flush=True)
    vm.close()
