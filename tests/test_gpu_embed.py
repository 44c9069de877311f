"""GPU tests of the embedding layer bench.py uses for the N-GPU launch: the in-process VM (libten4.so), the zero-copy
gradient slab, the gradient hook, and the library-owned RCCL communicator (world size 1 on the single test GPU)."""
import ctypes
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SRC = """0 trace
8 12 12 1 nn.model 0.5 4 conv2d 2 maxpool relu flatten 16 linear relu 10 linear softmax constant net
8 12 12 1 tensor rand constant img
: hot ( T -- T ) 8 0 do 1 i 10 * i 3 * 10 mod + t! loop ;
80 vector zeros hot 8 1 10 1 reshape4 constant lbl
: fb ( N -- N ) img forward lbl backprop ;
net fb
"""


def _num(txt, label):
    m = re.search(re.escape(label) + r"\s+([-+0-9.eE]+)", txt)
    assert m, (label, txt)
    return float(m.group(1))


@pytest.fixture(scope="module")
def vm():
    from tensorforth_amd.vm import VM
    v = VM(device=0, seed=7)
    out = v.eval(SRC)
    assert "?" not in out.replace("-> ok", ""), out
    yield v
    v.close()


def test_grad_slab_is_a_zero_copy_view_of_all_parameter_gradients(vm):
    import torch
    slab = vm.grad_slab()
    assert slab.is_cuda and slab.dtype == torch.float32 and slab.numel() >= 4 * 9 + 4 + 16 * 144 + 16 + 160 + 10
    txt = vm.eval('." s0 " 0 nn.dw sum . drop ." s1 " 0 nn.db sum . drop ." s2 " 4 nn.dw sum . drop ." s3 " 4 nn.db sum . drop '
                  '." s4 " 6 nn.dw sum . drop ." s5 " 6 nn.db sum . drop')
    want = sum(_num(txt, "s%d" % i) for i in range(6))
    torch.cuda.synchronize()
    got = float(slab.double().sum().item())                 # padding between tensors is zero
    assert abs(got - want) <= 2e-4 * max(1.0, abs(want)), (got, want)
    slab.zero_(); torch.cuda.synchronize()                  # writes through the view reach the VM's tensors
    assert _num(vm.eval('." z " 4 nn.dw sum . drop'), "z") == 0.0
    vm.eval("fb")                                           # restore gradients for the following tests


def test_gradient_hook_reports_tail_first_and_covers_the_slab(vm):
    slab = vm.grad_slab()
    calls = []
    vm.set_grad_hook(lambda layer, off, n: calls.append((layer, off, n)))
    vm.eval("fb")
    vm.set_grad_hook(None)
    assert [c[0] for c in calls] == [6, 4, 0]               # parameter layers, last first
    offs = [c[1] for c in calls]
    assert offs == sorted(offs, reverse=True) and offs[-1] == 0
    ends = [o + n for _, o, n in calls]
    assert ends[0] == slab.numel() and all(ends[i + 1] == calls[i][1] for i in range(len(calls) - 1))


def test_library_owned_communicator_world_size_one(vm, t4k):
    import torch
    lib = t4k.lib
    raw = (ctypes.c_ubyte * 128)()
    assert lib.t4k_comm_unique_id(raw) == 0, lib.t4k_last_error()
    assert any(raw), "empty communicator id"
    assert lib.t4k_comm_init(raw, 0, 1) == 0, lib.t4k_last_error()
    try:
        assert lib.t4k_comm_world() == 1 and lib.t4k_comm_rank() == 0
        x = torch.arange(1000, dtype=torch.float32, device="cuda")
        assert lib.t4k_allreduce_sum(x.data_ptr(), x.numel(), None) == 0
        lib.t4k_sync(None)
        assert torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float32))
        # with a communicator attached the VM sums its slab inside the optimizer word; one rank => same update as without
        before = _num(vm.eval('." w " 4 nn.w sum . drop'), "w")
        vm.eval("fb 0.01 0.0 nn.sgd")
        after = _num(vm.eval('." w " 4 nn.w sum . drop'), "w")
        assert after != before and np.isfinite(after)
    finally:
        lib.t4k_comm_destroy()
    assert lib.t4k_comm_world() == 0


_DP_SCRIPT = r'''
import ctypes, os, sys
sys.path.insert(0, os.environ["T4_ROOT"])
from tensorforth_amd.vm import VM
from tensorforth_amd import lib as t4lib
SRC = """0 trace
16 12 12 1 nn.model 0.5 6 conv2d 2 maxpool relu 0.5 8 conv2d relu flatten 300 linear relu 10 linear softmax constant net
16 12 12 1 tensor rand constant img
: hot ( T -- T ) 16 0 do 1 i 10 * i 3 * 10 mod + t! loop ;
160 vector zeros hot 16 1 10 1 reshape4 constant lbl
: step ( N -- N ) img forward lbl backprop 0.05 0.0 nn.sgd ;
: acc2 ( N -- N ) img forward lbl backprop img forward lbl backprop 0.05 0.0 nn.sgd ;
net
"""
v = VM(device=0, seed=11)
out = v.eval(SRC)                                        # (the first eval initialises the device library)
assert "?" not in out.replace("-> ok", ""), out
k = t4lib.load()
if os.environ.get("WITH_COMM") == "1":
    raw = (ctypes.c_ubyte * 128)()
    assert k.lib.t4k_comm_unique_id(raw) == 0 and k.lib.t4k_comm_init(raw, 0, 1) == 0, k.lib.t4k_last_error()
v.eval("step step step acc2 step")
print(v.eval('." W6 " 6 nn.w sum . drop ." W8 " 8 nn.w sum . drop ." W0 " 0 nn.w sum . drop ." W3 " 3 nn.w sum . drop lbl loss.ce ." CE " .'))
'''


def test_overlapped_slab_reduction_equals_in_order_training(tmp_path):
    """T4_DP_OVERLAP=2 forces the early-bucket path (event -> communication stream -> RCCL -> join) with a one-rank communicator:
    three plain steps, a two-backprop accumulation step (the 1/world rescale path) and another step must leave exactly the weights
    and loss of the same script without a communicator."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    f = tmp_path / "dp.py"; f.write_text(_DP_SCRIPT)
    outs = []
    for comm, ov, bucket in (("0", "1", "16384"), ("1", "2", "1024"), ("1", "2", "100000000"), ("1", "0", "16384")):
        env = dict(os.environ, T4_ROOT=root, WITH_COMM=comm, T4_DP_OVERLAP=ov, T4_DP_BUCKET=bucket, T4_DP_TRACE="1")
        r = subprocess.run([sys.executable, str(f)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        early = r.stderr.count("dp: early all-reduce"); final = r.stderr.count("dp: final all-reduce")
        if comm == "0": assert early == 0 and final == 0
        elif ov == "2" and bucket == "1024": assert early >= 8 and final == 5, r.stderr      # several buckets in each of the 4 one-backprop steps
        else: assert early == 0 and final == 5, r.stderr                                     # bucket never fills / overlap off: one in-order reduction per step
        line = [l for l in r.stdout.splitlines() if "W6" in l][-1]
        outs.append(line[line.index("W6"):])
    assert outs[1] == outs[0] and outs[2] == outs[0] and outs[3] == outs[0], outs
    assert all(np.isfinite(_num(outs[0], lab)) for lab in ("W6", "W8", "W0", "W3", "CE"))


def test_reseeding_a_vm_restarts_its_own_stream():
    """ten4_rand_reseed (include/ten4.h): the embedded VM's Philox stream is (seed, position) of the VM, installed by every ten4_eval - a host
    that wants another stream says so to the VM (a t4k_rand_init between evals would be overridden)."""
    import numpy as np
    from tensorforth_amd.vm import VM
    vm = VM(device=0, seed=77)
    a = vm.fetch("64 vector rand"); vm.eval("drop\n")
    b = vm.fetch("64 vector rand"); vm.eval("drop\n")
    assert not np.array_equal(a, b)                      # the stream moved on
    vm.rand_reseed(77)
    assert np.array_equal(vm.fetch("64 vector rand"), a); vm.eval("drop\n")     # same seed, position 0 again
    vm.rand_reseed(78)
    assert not np.array_equal(vm.fetch("64 vector rand"), a); vm.eval("drop\n")
    vm.close()
