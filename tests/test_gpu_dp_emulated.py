"""GPU: BASELINE config #5's arithmetic through the PRODUCT VM, with the ranks emulated one after the other on one GPU.

Two "ranks" (two VM instances, same seed => identical replicas) each train on their half of a 16-image batch with the LeNet net of
bench.py (both dropouts on): shard (r, 2) set with t4k_rand_set_shard, same stream position, the two gradient slabs summed on the
host and written back (what the RCCL all-reduce(SUM) does between `backprop` and `nn.sgd`), then the optimizer word.  A third VM
trains on the whole 16-image batch.  Everything the scripts print must agree: the weights after the step (SUM over shards == large
batch, raw batch-sum gradients, quirk a-19), the dropout-mask sums (rank 0 + rank 1 == whole batch: masks keyed by sample) and the
forward outputs.  Only the transport is emulated - kernels, fused launch plan, slab layout and host orchestration are the product's."""
import ctypes

import numpy as np
import pytest

from vm_util import numbers_after

pytestmark = pytest.mark.gpu

NET = "0.5 10 conv2d 2 maxpool relu 0.5 20 conv2d 0.5 dropout 2 maxpool relu flatten 100 linear 0.5 dropout 10 linear softmax"
REPORT = ('." mask_conv " 4 nn.ex sum . drop ." mask_lin " 9 nn.ex sum . drop ." w0 " 0 nn.w . ." b3 " 3 nn.b . ." w8 " 8 nn.w sum . drop '
          '." b8 " 8 nn.b sum . drop ." w10 " 10 nn.w . ." b10 " 10 nn.b .\n')


def _build(VM, k, seed, n, row0, rows_total):
    """VM with the net, its rows [row0, row0+n) of the whole batch's image draw, labels by global row index"""
    vm = VM(device=0, seed=seed)
    vm.eval("0 trace\n%d 28 28 1 nn.model %s constant net\n" % (n, NET))
    off0 = k.lib.t4k_rand_offset()
    k.call("t4k_rand_set_offset", off0 + row0 * 784)
    vm.eval("%d 28 28 1 tensor rand constant img\n" % n)
    k.call("t4k_rand_set_offset", off0 + rows_total * 784)
    out = vm.eval(": hot ( T -- T ) %d 0 do 1 i 10 * i %d + 7 * 10 mod + t! loop ;\n"
                  "%d vector zeros hot %d 1 10 1 reshape4 constant lbl\n"
                  ": fb ( N -- N ) img forward lbl backprop ;\n: opt ( N -- N ) 0.01 0.0 nn.sgd ;\n" % (n, row0, n * 10, n))
    assert "?" not in out.replace("-> ok", ""), out
    return vm, k.lib.t4k_rand_offset()


def test_two_emulated_ranks_equal_the_whole_batch():
    import torch
    from tensorforth_amd import lib as t4lib
    from tensorforth_amd.vm import VM
    k = t4lib.load()
    N, seed = 16, 77
    whole, off_w = _build(VM, k, seed, N, 0, N)
    r0, off_0 = _build(VM, k, seed, N // 2, 0, N)
    r1, off_1 = _build(VM, k, seed, N // 2, N // 2, N)
    assert off_w == off_0 == off_1                              # identical replicas: every VM is at the same stream position
    try:
        for step in range(2):
            k.call("t4k_rand_set_shard", 0, 1); k.call("t4k_rand_set_offset", off_w)
            whole.eval("net fb opt drop\n"); end = k.lib.t4k_rand_offset()
            slabs = []
            for r, vm in enumerate((r0, r1)):
                k.call("t4k_rand_set_shard", r, 2); k.call("t4k_rand_set_offset", off_w)
                vm.eval("net fb drop\n")
                assert k.lib.t4k_rand_offset() == end, "a rank moves the stream by the whole batch's draws"
                slabs.append(vm.grad_slab())
            torch.cuda.synchronize()
            total = slabs[0] + slabs[1]                         # the all-reduce(SUM) of the gradient slab
            for s_ in slabs:
                s_.copy_(total)
            torch.cuda.synchronize()
            for vm in (r0, r1):
                vm.eval("net opt drop\n")
            off_w = end
        k.call("t4k_rand_set_shard", 0, 1)
        ow = whole.eval("net " + REPORT + "drop\n")
        o0 = r0.eval("net " + REPORT + "drop\n"); o1 = r1.eval("net " + REPORT + "drop\n")
    finally:
        k.call("t4k_rand_set_shard", 0, 1)
        for vm in (whole, r0, r1):
            vm.close()
    for lab in ("mask_conv", "mask_lin"):                       # masks of the last step: keyed by sample => the shards partition the whole batch's mask
        assert numbers_after(o0, lab, 1)[0] + numbers_after(o1, lab, 1)[0] == numbers_after(ow, lab, 1)[0], lab
    for lab, cnt in (("w0", 90), ("b3", 20), ("w8", 1), ("b8", 1), ("w10", 60), ("b10", 10)):
        a, b, w = (np.array(numbers_after(o, lab, cnt)) for o in (o0, o1, ow))
        assert np.array_equal(a, b), "replicas diverged at " + lab
        np.testing.assert_allclose(a, w, rtol=2e-4, atol=2.5e-4, err_msg=lab)
