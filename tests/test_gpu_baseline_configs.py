"""GPU: BASELINE configs #2, #3 and #4 at FULL size through the product host.

* config #3 exactly as bench.py runs it - the in-process VM (libten4.so), `nn_f` LeNet net, batch 128, both dropouts on,
  `forward backprop 0.01 nn.sgd` in a compiled loop with the fused 13-launch plan - against the LIVE oracle VM (`oracle/ten4_oracle`,
  CPU) on seeds the golden files were not made with: every parameter tensor, both dropout masks, the loss, the gradients.
* config #4: the t4_40b GAN nets at N = 256, two `train_d train_g` rounds, same comparison.
* config #2: the `matmul` / `@` WORD on 1024 x 1024 integer-valued operands (entries in {-2..2}: every product and partial sum is
  exact in fp32 whatever the summation order), compared bit for bit with numpy's integer product - no oracle in the loop.
(The golden-file versions of the same scripts run in test_vm_scripts.py, in every execution-engine variant.)"""
import os

import numpy as np
import pytest

from vm_util import SCRIPTS, TEN4, TEN4_ORACLE, compare, numbers_after, run_vm, tokens

pytestmark = pytest.mark.gpu


def _src(name):
    with open(os.path.join(SCRIPTS, name + ".4th")) as f:
        return f.read()


def _inproc(src, seed):
    from tensorforth_amd.vm import VM
    vm = VM(device=0, seed=seed)
    try:
        return "tensorForth v4.0\n" + vm.eval(src) + "\ntensorForth done.\n"
    finally:
        vm.close()


@pytest.mark.parametrize("seed", [7, 2024])
def test_config3_lenet128_inprocess_vm_vs_live_oracle_vm(seed):
    if not os.path.exists(TEN4_ORACLE):
        pytest.skip("oracle VM binary not shipped")
    want = run_vm(TEN4_ORACLE, os.path.join(SCRIPTS, "cfg3_lenet128.4th"), seed=seed)
    got = _inproc(_src("cfg3_lenet128"), seed)
    bad = compare(got, want, rtol=3e-4, atol=5e-4)
    assert bad == [], "\n".join(bad)
    # dropout masks come from the same Philox stream: their sums (0/1 entries) are integers and must be identical
    for label in ("mask_conv", "mask_lin"):
        assert numbers_after(got, label, 1) == numbers_after(want, label, 1)
    # and the stand-alone binary prints what the embedded VM prints (same sources, same library)
    assert tokens(run_vm(TEN4, os.path.join(SCRIPTS, "cfg3_lenet128.4th"), seed=seed)) == tokens(got)


def test_config4_gan256_vs_live_oracle_vm():
    if not os.path.exists(TEN4_ORACLE):
        pytest.skip("oracle VM binary not shipped")
    seed = 31
    want = run_vm(TEN4_ORACLE, os.path.join(SCRIPTS, "cfg4_gan256.4th"), seed=seed, timeout=600)
    got = run_vm(TEN4, os.path.join(SCRIPTS, "cfg4_gan256.4th"), seed=seed)
    bad = compare(got, want, rtol=1e-3, atol=2e-3)          # post-Adam weights: eps = 1e-6 outside the sqrt amplifies 1-ulp gradient differences
    assert bad == [], "\n".join(bad)
    for label in ("d_mask2", "d_mask5"):
        assert numbers_after(got, label, 1) == numbers_after(want, label, 1)


def test_config2_matmul_word_1024_is_exact_on_integer_operands():
    rng = np.random.default_rng(1024)
    A = rng.integers(-2, 3, (1024, 1024)); B = rng.integers(-2, 3, (1024, 1024))
    C = A @ B                                                  # int64: the truth
    lit = lambda m: " ".join(str(int(v)) for v in m.ravel())
    picks = [0, 1023, 1024 * 517 + 33, 1024 * 1023 + 1023, 1024 * 64 + 63, 1024 * 63 + 64]
    src = ("0 trace\n1024 1024 matrix{ %s } constant ma\n1024 1024 matrix{ %s } constant mb\n" % (lit(A), lit(B)) +
           "ma mb matmul constant mc\nma mb @ constant md\n" +
           'mc max ." cmax " . drop mc min ." cmin " . drop\n' +
           "".join('mc %d t@ ." e%d " . drop md %d t@ ." f%d " . drop\n' % (i, k, i, k) for k, i in enumerate(picks)) +
           'mc ." C " .\n')
    out = _inproc(src, 1)
    assert numbers_after(out, "cmax", 1) == [float(C.max())] and numbers_after(out, "cmin", 1) == [float(C.min())]
    for k, i in enumerate(picks):
        assert numbers_after(out, "e%d" % k, 1) == [float(C.ravel()[i])], (k, i)
        assert numbers_after(out, "f%d" % k, 1) == [float(C.ravel()[i])], (k, i)
    # the printer elides to the first / last three rows and columns (aio_tensor.cpp:141-226 restated in host/printer.cpp)
    corner = C[np.ix_([0, 1, 2, 1021, 1022, 1023], [0, 1, 2, 1021, 1022, 1023])].astype(np.float64).ravel()
    assert numbers_after(out, "C", 36) == list(corner)
