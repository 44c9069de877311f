"""GPU: the library without its inter-workgroup gates (t4k_gates_enable(0); what it switches to by itself after a wait timed out, runtime.hip spin_check -
VERDICT r4 weak #11: a partitioned or shared device must cost speed, not correctness).  The LeNet step of BASELINE config #3 and a GAN-shaped linear
stack are trained both ways in processes of their own: same parameters to 1e-4 (other kernels, other summation orders), more launches per step, and no
kernel of the ungated run may be one of the gated kinds."""
import os
import subprocess
import sys

import numpy as np
import pytest

from vm_util import ROOT, rel_err

pytestmark = pytest.mark.gpu

_WORKER = r'''
import ctypes, os, sys
sys.path.insert(0, os.environ["T4_ROOT"]); sys.path.insert(0, os.path.join(os.environ["T4_ROOT"], "tests"))
import numpy as np
from tensorforth_amd import lib as t4lib
from tensorforth_amd.vm import VM
from lenet_parity import PARAMS, _get, _setup
k = t4lib.load()
vm = VM(device=0, seed=77)
if os.environ["GATES"] == "0":
    assert k.lib.t4k_gates_enable(0) == 0 and k.lib.t4k_gates_enabled() == 0
_setup(vm, 128, 0, 128)
k.lib.t4k_launch_count.restype = ctypes.c_ulonglong
vm.eval("net fw bw opt drop\n")
l0 = k.lib.t4k_launch_count()
for _ in range(3):
    vm.eval("net fw bw opt drop\n")
per = (k.lib.t4k_launch_count() - l0) / 3.0
assert k.lib.t4k_sync(None) == 0, k.lib.t4k_last_error()
out = {n: _get(vm, e) for n, e in PARAMS}
# a linear stack in the GAN's shapes (sliver GEMMs, dual dW || dX launches with the in-place arrival gate)
vm.eval("256 1 128 1 nn.model 256 linear 0.2 leakyrelu 512 linear 0.2 leakyrelu 784 linear tanh constant gen\n256 1 128 1 tensor randn constant z\n256 1 784 1 tensor rand constant tgt\n")
for _ in range(3):
    vm.eval("gen z forward tgt backprop 0.0004 0.5 nn.adam drop\n")
assert k.lib.t4k_sync(None) == 0, k.lib.t4k_last_error()
for L in (0, 2, 4):
    a = vm.fetch("gen %d nn.w" % L); vm.eval("drop drop"); out["g%d" % L] = a
np.savez(os.environ["OUT"], launches=per, **out)
'''


def _run(tmp_path, gates):
    out = str(tmp_path / ("g%s.npz" % gates))
    r = subprocess.run([sys.executable, "-c", _WORKER], env=dict(os.environ, T4_ROOT=ROOT, GATES=gates, OUT=out), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return np.load(out)


def test_ungated_library_trains_the_same_model_with_more_launches(tmp_path):
    on, off = _run(tmp_path, "1"), _run(tmp_path, "0")
    assert float(on["launches"]) <= 5.01, float(on["launches"])
    assert float(off["launches"]) > float(on["launches"]) + 0.99, (float(on["launches"]), float(off["launches"]))   # the head leaves the stack's forward, the dual products split
    for n in on.files:
        if n == "launches":
            continue
        if n.startswith("g"):                                 # Adam: sign-sized steps on rounding-level gradient elements (tests/test_gpu_baseline_configs.py _adam_check) - compare in bulk
            d = np.abs(on[n].astype(np.float64) - off[n])
            assert (d > 1e-4 * np.abs(off[n]).max()).mean() <= 2e-3, (n, float((d > 1e-4 * np.abs(off[n]).max()).mean()))
        else:
            assert rel_err(on[n], off[n]) <= 1e-4, (n, rel_err(on[n], off[n]))
