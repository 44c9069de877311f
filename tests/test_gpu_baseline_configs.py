"""GPU: BASELINE configs #2, #3 and #4 at FULL size through the product host.

Configs #3 and #4 are compared on FULL fp32 tensors (ten4_fetch on the product VM, the same call on the oracle VM in its worker
process) at the north_star bar - 1e-4 relative per tensor, every element; the printed text is compared as well, to one unit of the
printer's last place.  Config #5 at its own size: tests/test_gpu_config5_full.py.

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

from vm_util import SCRIPTS, TEN4, TEN4_ORACLE, OracleVM, check_tensor, compare, numbers_after, rel_err, run_vm, tokens

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


TOL = 1e-4                                                    # north_star: outputs within 1e-4 relative of the reference
# the printer shows tensors with 4 decimals (aio_tensor.cpp:141-226): one unit of the last printed place + the fp32 bar
TEXT_RTOL, TEXT_ATOL = 1e-4, 1.01e-4


def _body(name, until):
    """the script's source up to (not including) the first line that starts with `until`"""
    out = []
    for line in _src(name).splitlines():
        if line.startswith(until):
            break
        out.append(line)
    return "\n".join(out) + "\n"


def _pair(seed):
    from tensorforth_amd.vm import VM
    return VM(device=0, seed=seed), OracleVM(seed=seed)


def _fetch(vm, model, expr):
    a = vm.fetch("%s %s" % (model, expr)); vm.eval("drop drop")
    return a


@pytest.mark.parametrize("seed", [7, 2024])
def test_config3_lenet128_full_tensors_vs_oracle_vm(seed):
    """config #3 exactly as bench.py runs it (in-process VM, the sample-resident conv stack + fused launch plan, both dropouts on):
    three training steps, each compared with the oracle VM on full fp32 tensors (tests/lenet_parity.py: every gradient, dX, both
    masks, the post-SGD parameters at 1e-4; a max-pool arg-max tie - seed 7 has one - is verified to BE a tie and the conv gradients
    are then held to float64 on the product's own operands).  Each step starts from the oracle's parameters."""
    import lenet_parity as lp
    g, o = _pair(seed)
    try:
        for vm in (g, o):
            assert lp._setup(vm, 128, 0, 128) > 0
        img = g.fetch("img"); g.eval("drop")
        assert np.array_equal(o.fetch("img"), img); o.eval("drop")
        ties = 0
        for step in range(3):
            ties += len(lp.step_vs_oracle(g, o, img, step))
        assert ties <= 4
    finally:
        g.close(); o.close()


@pytest.mark.parametrize("seed", [2024])          # (seed 7 meets a max-pool arg-max tie in step 1: printed sums then differ by 2e-4 - see the full-tensor test)
def test_config3_lenet128_printed_text_vs_live_oracle_vm(seed):
    if not os.path.exists(TEN4_ORACLE):
        pytest.skip("oracle VM binary not shipped")
    want = run_vm(TEN4_ORACLE, os.path.join(SCRIPTS, "cfg3_lenet128.4th"), seed=seed)
    got = _inproc(_src("cfg3_lenet128"), seed)
    bad = compare(got, want, rtol=TEXT_RTOL, atol=TEXT_ATOL)
    assert bad == [], "\n".join(bad)
    # dropout masks come from the same Philox stream: their sums (0/1 entries) are integers and must be identical
    for label in ("mask_conv", "mask_lin"):
        assert numbers_after(got, label, 1) == numbers_after(want, label, 1)
    # and the stand-alone binary prints what the embedded VM prints (same sources, same library)
    assert tokens(run_vm(TEN4, os.path.join(SCRIPTS, "cfg3_lenet128.4th"), seed=seed)) == tokens(got)


ADAM_B1, ADAM_B2, ADAM_EPS = 0.5, 0.999, 1e-6               # the script's beta1, the word's default beta2 (netvm.cpp:393-399), DU_EPS


def _adam_check(name, got, want, lr, g_first=None, steps=1):
    """Post-Adam weights.  The reference's update has no bias correction and adds eps OUTSIDE the root (nmath.cu:438-454):
    first step s(g) = lr (1-b1) g / (sqrt(1-b2) |g| + eps) - a steep, monotone function of g near zero (slope lr (1-b1) / eps = 50..200),
    flat (+-15.8 lr) elsewhere.  A gradient that is right to 1e-4 of its tensor therefore gives a weight that is right to
        1e-4 max|w|  +  |s(g + eta) - s(g - eta)|,   eta = 1e-4 max|g|,
    which is the bar used for the FIRST step, element by element (g = the oracle's gradient).  For later steps (m, v carry state) the
    same effect is bounded in bulk: 99.9 % of the elements within 1e-4 relative, the rest within the largest movement `steps` updates
    can make (15.8 lr each)."""
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    d = np.abs(got - want); scale = max(1e-30, np.max(np.abs(want)))
    if g_first is not None:
        g = np.asarray(g_first, np.float64).reshape(want.shape)
        s_ = lambda x: lr * (1 - ADAM_B1) * x / (np.sqrt(1 - ADAM_B2) * np.abs(x) + ADAM_EPS)
        eta = TOL * np.max(np.abs(g))
        bound = TOL * scale + np.abs(s_(g + eta) - s_(g - eta))
        worst = np.max(d / bound)
        assert worst <= 1.0, "%s: |d| exceeds the propagated 1e-4 gradient bar by x%.3g" % (name, worst)
        return
    loose = d > TOL * scale
    assert loose.mean() <= 1e-3, "%s: %.3g of the elements beyond 1e-4 relative" % (name, loose.mean())
    step_max = lr * (1 - ADAM_B1) / np.sqrt(1 - ADAM_B2)
    assert d.max() <= 2.0 * steps * step_max, "%s: |d| = %.3g exceeds what %d Adam steps can move" % (name, d.max(), steps)


# Element-aware bar for the GAN's gradient tensors (vm_util.check_tensor; the tensor-norm bar of 1e-4 holds for EVERY element regardless):
# D's gradients accumulate a REAL and a FAKE backprop whose contributions largely cancel, G's have passed backwards through six GEMMs - an
# element below a hundredth of its tensor's largest has no 1e-4 accuracy of its own on either side.  Measured over three seeds, floor 1e-3:
# 15 of 131 072 (D 3 dW), 3 of 256 (D 6 dW), 45 of 401 408 (G 4 dW) elements beyond the bar; floor 1e-2: 420 of 200 704 (0.2 %) of the frozen
# discriminator's dX on the worst seed, a handful elsewhere.  The LeNet configs (#3, #5) hold 99.99 % (tests/lenet_parity.py); here the bar is 99 %.
# Round 5 ("tighten to floor 1e-3 / 99.9 % if the data allow"): they allow it for the DISCRIMINATOR's parameter gradients (0.011 % of the large
# tensors, 3 elements of the 256-element one: min_outliers covers tensors too small for a percentage) - those are now held to the LeNet floor.  They do not for
# the GENERATOR's (gradients that have passed backwards through all six products): at floor 1e-3 its first layer shows 38 to 386 of 32 768 (0.1 - 1.2 %) depending
# on the seed, its last layer up to 1 202 of 401 408 (0.3 %), its 784 bias gradients 6 - nor for the frozen discriminator's dX; those keep round 4's bar AGAINST THE ORACLE.
# Round 6 settles whose rounding that is: the generator phase is ALSO computed in float64 on the oracle VM's operands
# (_gan_train_g_vs_float64) and both fp32 sides are measured against it at floor 1e-3 - product 0 - 6 elements of 32 768 beyond the bar in G's first layer and none in
# any other tensor, the oracle (the reference's sequential fp32 sums) 30 - 90 there and in the frozen D's dX, rms error of the product 3 - 4 x smaller in every tensor.
# The product is held to floor 1e-3 / 99.95 % against float64 and to "not farther from exact than the oracle"; train_g now runs on ONE set of D parameters in both VMs
# (Adam's +-lr steps on rounding-level gradients had left them 2 lr apart in single elements - two different functions).
GAN_ELEM_D = dict(floor=1e-3, elem_min=0.999, min_outliers=4)
GAN_ELEM = dict(floor=1e-2, elem_min=0.99, min_outliers=2)
GAN_ELEM_DX = GAN_ELEM


def _gan_train_g_vs_float64(tag, g, o):
    """The generator phase in float64 on the oracle VM's operands (its parameters, Z, its dropout and leakyrelu masks): forward G -> frozen D, `out - REAL`
    backwards through D (no parameter gradients) and G, as Model::backprop does it (sigmoid passes through, tanh multiplies by 1 - y^2: backprop.cu:97-131).
    Both fp32 sides are measured against it: the product may not be farther from the exact result than the reference's arithmetic is."""
    f8 = lambda a: np.asarray(a, dtype=np.float64)
    Z = f8(o.fetch("Z")).reshape(256, 128); o.eval("drop")
    Wg = {L: f8(_fetch(o, "G", "%d nn.w" % L)).reshape(-1, e1) for L, e1 in ((0, 128), (2, 256), (4, 512))}
    Bg = {L: f8(_fetch(o, "G", "%d nn.b" % L)).ravel() for L in (0, 2, 4)}
    Wd = {L: f8(_fetch(o, "D", "%d nn.w" % L)).reshape(-1, e1) for L, e1 in ((0, 784), (3, 512), (6, 256))}
    Bd = {L: f8(_fetch(o, "D", "%d nn.b" % L)).ravel() for L in (0, 3, 6)}
    mg = {L: f8(_fetch(o, "G", "%d nn.ex" % L)).reshape(256, -1) for L in (1, 3)}          # leakyrelu derivative 1 / 0.2 = the factor of the forward as well
    md = {L: f8(_fetch(o, "D", "%d nn.ex" % L)).reshape(256, -1) for L in (1, 2, 4, 5)}    # 1, 4 leakyrelu; 2, 5 dropout (0 / 1, no rescale)
    a1 = (Z @ Wg[0].T + Bg[0]) * mg[1]
    a2 = (a1 @ Wg[2].T + Bg[2]) * mg[3]
    f = np.tanh(a2 @ Wg[4].T + Bg[4])
    e1 = (f @ Wd[0].T + Bd[0]) * md[1] * md[2]
    e2 = (e1 @ Wd[3].T + Bd[3]) * md[4] * md[5]
    out = 1.0 / (1.0 + np.exp(-(e2 @ Wd[6].T + Bd[6])))
    g3 = out - 1.0
    gd2 = (g3 @ Wd[6]) * md[5] * md[4]
    gd1 = (gd2 @ Wd[3]) * md[2] * md[1]
    gf = gd1 @ Wd[0]                                            # dX of the frozen discriminator = the generator's target gradient
    gh3 = gf * (1.0 - f * f)
    gh2 = (gh3 @ Wg[4]) * mg[3]
    gh1 = (gh2 @ Wg[2]) * mg[1]
    want = {("D", "0 n@"): gf, ("G", "4 nn.dw"): gh3.T @ a2, ("G", "4 nn.db"): gh3.sum(0), ("G", "2 nn.dw"): gh2.T @ a1, ("G", "2 nn.db"): gh2.sum(0),
            ("G", "0 nn.dw"): gh1.T @ Z, ("G", "0 nn.db"): gh1.sum(0)}
    for (m, e), w in want.items():
        w = w.ravel(); pg = f8(_fetch(g, m, e)).ravel(); po = f8(_fetch(o, m, e)).ravel()
        bar = TOL * np.maximum(np.abs(w), 1e-3 * np.abs(w).max())
        ng, no = int((np.abs(pg - w) > bar).sum()), int((np.abs(po - w) > bar).sum())
        rg, ro = float(np.sqrt(np.mean((pg - w) ** 2))), float(np.sqrt(np.mean((po - w) ** 2)))
        GAN_F64_LOG.append("%s %s %s: beyond the element bar (floor 1e-3) product %d oracle %d of %d; rms error product %.3g oracle %.3g; max %.3g / %.3g of max|ref|"
                           % (tag, m, e, ng, no, w.size, rg, ro, np.abs(pg - w).max() / np.abs(w).max(), np.abs(po - w).max() / np.abs(w).max()))
        name = "%s %s %s vs float64" % (tag, m, e)
        assert np.abs(pg - w).max() <= TOL * np.abs(w).max(), "%s: %.3g of max|ref|" % (name, np.abs(pg - w).max() / np.abs(w).max())
        # measured, 3 seeds x 2 rounds: product 0 - 6 elements beyond the bar (G's first layer, 32 768 elements; 0 everywhere else), oracle 30 - 90 there and in
        # the frozen D's dX; rms error of the product 3 - 4 x below the oracle's in every tensor
        assert ng <= max(2, int(5e-4 * w.size)), "%s: %d of %d elements beyond 1e-4 of their own magnitude (floor 1e-3 max|ref|)" % (name, ng, w.size)
        assert ng <= no + 2 and rg <= ro, "%s: the product is farther from the exact result than the oracle (%d vs %d elements, rms %.3g vs %.3g)" % (name, ng, no, rg, ro)


GAN_F64_LOG = []


def _gan_two_rounds(seed):
    """Two `train_d train_g` rounds of the product VM against the oracle VM, phase by phase, on full tensors.  Returns the list of
    activation-kink flips met on the way (empty: none).  A pre-activation within rounding of the leakyrelu kink makes the fp32 summation
    order decide the branch - in the reference as much as here; such a flip is VERIFIED to be at the kink (|x| <= 1e-5 of the tensor's
    scale), the gradients of that phase then describe two equally valid trajectories, so the product VM takes over the oracle's gradient
    tensors (ten4_store) and the run CONTINUES: every seed goes through both rounds."""
    g, o = _pair(seed)
    try:
        src = _body("cfg4_gan256", "D 2 rounds")
        for vm in (g, o):
            out = vm.eval(src)
            assert "?" not in out.replace("-> ok", ""), out
        for m, e in (("D", "0 nn.w"), ("D", "3 nn.w"), ("D", "6 nn.w"), ("G", "0 nn.w"), ("G", "2 nn.w"), ("G", "4 nn.w")):
            assert np.array_equal(_fetch(g, m, e), _fetch(o, m, e)), "initial weights are the same Philox draw: " + m + " " + e
        flips = []
        og = {}                                                  # the oracle's gradients in front of each optimizer call

        def kinks(tag):
            """leakyrelu derivative masks (1 / 0.2) must agree; where they do not, the pre-activation must sit AT the kink"""
            n = 0
            for m, acts in (("D", (1, 4)), ("G", (1, 3))):
                for L in acts:
                    a, b = _fetch(g, m, "%d nn.ex" % L), _fetch(o, m, "%d nn.ex" % L)
                    if not np.array_equal(a, b):
                        x = _fetch(o, m, "%d n@" % (L + 1))            # the activation's output: |y| = |x| or 0.2 |x|
                        bad = np.argwhere(a != b)
                        for i_ in bad:
                            assert abs(x[tuple(i_)]) <= 1e-5 * np.abs(x).max(), "%s %s layer %d: masks differ AWAY from the kink at %s" % (tag, m, L, i_)
                        assert len(bad) <= 4, "%s %s layer %d: %d mask elements differ" % (tag, m, L, len(bad))
                        flips.append("%s %s layer %d: %d element(s) at the leakyrelu kink" % (tag, m, L, len(bad))); n += len(bad)
            return n

        def adopt(model, exprs):
            """the product VM continues from the oracle's tensors"""
            for e in exprs:
                g.store(_fetch(o, model, e), "%s %s" % (model, e)); g.eval("drop drop")
        DG = ("0 nn.dw", "0 nn.db", "3 nn.dw", "3 nn.db", "6 nn.dw", "6 nn.db")
        GG = ("4 nn.dw", "4 nn.db", "2 nn.dw", "2 nn.db", "0 nn.dw", "0 nn.db")
        for rnd in (1, 2):
            # one round by hand: forward values and raw gradients are well conditioned -> 1e-4 (tensor norm AND element-aware); dropout masks bit-exact
            for vm in (g, o):
                vm.eval("D 1 trainable real forward REAL backprop F forward FAKE backprop\n")
            flipped = kinks("round %d train_d" % rnd)
            for e in ("2 nn.ex", "5 nn.ex"):
                assert np.array_equal(_fetch(g, "D", e), _fetch(o, "D", e)), "D " + e
            for e in DG:
                b = _fetch(o, "D", e); og[("D", e.replace("nn.d", "nn."))] = b
                if not flipped:
                    check_tensor("round %d D %s (two accumulated backprops)" % (rnd, e), _fetch(g, "D", e), b, TOL, **GAN_ELEM_D)
            if flipped:
                adopt("D", DG)
            for vm in (g, o):
                vm.eval("0.0001 0.5 nn.adam\n")
            for e, lr in (("0 nn.w", 1e-4), ("0 nn.b", 1e-4), ("3 nn.w", 1e-4), ("6 nn.w", 1e-4), ("6 nn.b", 1e-4)):
                _adam_check("round %d D %s" % (rnd, e), _fetch(g, "D", e), _fetch(o, "D", e), lr, g_first=og[("D", e)] if rnd == 1 else None, steps=rnd)
            # train_g runs on ONE set of discriminator parameters (the oracle's): Adam's +-lr steps on rounding-level gradient elements leave the two
            # VMs with weights that differ by up to 2 lr in single elements, and the generator's gradients would then compare two different functions
            adopt("D", ["%d %s" % (L, kind) for L in (0, 3, 6) for kind in ("nn.w", "nn.b")])
            for vm in (g, o):
                vm.eval("0 trainable F forward REAL backprop 0 n@ G swap backprop\n")
            flipped2 = kinks("round %d train_g" % rnd)
            if not flipped2:
                _gan_train_g_vs_float64("round %d" % rnd, g, o)
            if not flipped2:
                # (floor 1e-2 from here on: these gradients have passed through the discriminator's three layers backwards and - for G's - the
                # generator's as well; an element below a hundredth of the tensor's largest carries the rounding of a six-GEMM chain on both sides)
                check_tensor("round %d dX of the frozen D" % rnd, _fetch(g, "D", "0 n@"), _fetch(o, "D", "0 n@"), TOL, **GAN_ELEM_DX)
            for e in GG:
                b = _fetch(o, "G", e); og[("G", e.replace("nn.d", "nn."))] = b
                if not flipped2:
                    check_tensor("round %d G %s" % (rnd, e), _fetch(g, "G", e), b, TOL, **GAN_ELEM)
            if flipped2:
                adopt("G", GG)
            for vm in (g, o):
                vm.eval("0.0004 0.5 nn.adam drop\n")
            assert g.rand_tell() == o.rand_tell()
            for m, e, lr in (("G", "0 nn.w", 4e-4), ("G", "2 nn.w", 4e-4), ("G", "4 nn.w", 4e-4), ("G", "4 nn.b", 4e-4)):
                _adam_check("round %d %s %s" % (rnd, m, e), _fetch(g, m, e), _fetch(o, m, e), lr, g_first=og[(m, e)] if rnd == 1 else None, steps=rnd)
            if rnd == 1:
                # Round 2 starts from ONE set of parameters (the oracle's, written into the product VM at full precision): the +-15.8 lr
                # steps Adam takes on rounding-level gradient elements would otherwise make round 2 a comparison of two trajectories
                for m, layers in (("D", (0, 3, 6)), ("G", (0, 2, 4))):
                    adopt(m, ["%d %s" % (L, kind) for L in layers for kind in ("nn.w", "nn.b")])
        for vm in (g, o):
            vm.eval("G Z forward\n")
        err = rel_err(_fetch(g, "G", "-1 n@"), _fetch(o, "G", "-1 n@"))
        assert err <= 2 * TOL, "generator output after two rounds (tanh of a 3-layer product of post-Adam weights): %.3g" % err
        return flips
    finally:
        g.close(); o.close()


def test_config4_gan256_full_tensors_vs_oracle_vm():
    """config #4: the t4_40b GAN nets at N = 256, two `train_d train_g` rounds (BCE, Adam beta1 = 0.5, dropout in D), every gradient
    tensor at 1e-4 relative (tensor norm and element-aware), post-Adam weights by _adam_check - ALL THREE seeds go through both rounds.
    Seed 31 meets a pre-activation within rounding of the leakyrelu kink in round 2 (|x| = 6e-7, one element of 131 072): the flip is
    verified to be AT the kink, the product VM takes over the oracle's gradients for that phase, and the run continues."""
    notes = {seed: _gan_two_rounds(seed) for seed in (31, 47, 2024)}
    print("\n".join(GAN_F64_LOG))
    assert sum(len(v) for v in notes.values()) <= 3, notes           # kinks are rare events: a handful over three seeds at most
    assert sum(1 for v in notes.values() if not v) >= 2, notes       # ... and most seeds never meet one


def test_config4_gan256_printed_text_vs_live_oracle_vm():
    if not os.path.exists(TEN4_ORACLE):
        pytest.skip("oracle VM binary not shipped")
    seed = 31
    want = run_vm(TEN4_ORACLE, os.path.join(SCRIPTS, "cfg4_gan256.4th"), seed=seed, timeout=600)
    got = run_vm(TEN4, os.path.join(SCRIPTS, "cfg4_gan256.4th"), seed=seed)
    # printed values here are SUMS over whole post-Adam weight tensors (up to 400 k elements of +-lr steps): a handful of sign-ambiguous
    # elements (see _adam_check) move a sum by a few 1e-4 - the full-tensor test above is the parity statement, this one checks the
    # script-level text (losses, mask sums, formats)
    bad = compare(got, want, rtol=1e-3, atol=2e-3)
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


# ---------------------------------------------------------------------------------------------------------------- t4_40a (the net config #4 names)
NET_40A = "0.5 10 conv2d 2 maxpool relu flatten 100 linear relu 10 linear softmax"       # examples/t4_40a.4th:10-13 (`nn_c`), N = 256, lr nn.adam (0.001, b1 = 0.9)
P40 = [("w0", "0 nn.w"), ("b0", "0 nn.b"), ("w4", "4 nn.w"), ("b4", "4 nn.b"), ("w6", "6 nn.w"), ("b6", "6 nn.b")]
G40 = [("dw0", "0 nn.dw"), ("db0", "0 nn.db"), ("dw4", "4 nn.dw"), ("db4", "4 nn.db"), ("dw6", "6 nn.dw"), ("db6", "6 nn.db")]


def _setup_40a(vm, n=256):
    out = vm.eval("0 trace\n%d 28 28 1 nn.model %s constant net\n%d 28 28 1 tensor rand constant img\n" % (n, NET_40A, n) +
                  ": hot ( T -- T ) %d 0 do 1 i 10 * i 7 * 10 mod + t! loop ;\n%d vector zeros hot %d 1 10 1 reshape4 constant lbl\n" % (n, n * 10, n) +
                  ": fw ( N -- N ) img forward ;\n: bw ( N -- N ) lbl backprop ;\n: opt ( N -- N ) 0.001 nn.adam ;\n")
    assert "?" not in out.replace("-> ok", ""), out


def test_t4_40a_net_at_its_own_size_three_adam_steps_vs_oracle_vm():
    """The net BASELINE config #4 literally names (examples/t4_40a.4th:10-13 + 27-31: `nn_c`, N = 256, `forward backprop lr nn.adam` with the
    word's defaults b1 = 0.9, b2 = 0.999) through the C++ VM - sample-resident conv stack with the classifier head in its launch (mid layer =
    relu), fused head backward, fold inside the Adam launch - against the oracle VM on FULL tensors: forward output, every gradient (1e-4,
    tensor norm and element-aware; a max-pool arg-max tie is verified and the filter gradient then held to float64 on each side's own operands),
    post-Adam parameters by the propagated-gradient rule of _adam_check (first step) / its bulk rule (later steps)."""
    import lenet_parity as lp
    global ADAM_B1
    g, o = _pair(77)
    b1_old = ADAM_B1
    try:
        ADAM_B1 = 0.9
        for vm in (g, o):
            _setup_40a(vm)
        img = g.fetch("img"); g.eval("drop")
        assert np.array_equal(o.fetch("img"), img); o.eval("drop")
        ties = 0
        for step in range(3):
            g.eval("net fw\n"); o.eval("net fw\n")
            c1o = lp._get(o, "1 n@")                                 # conv output the max-pool selects from
            check_tensor("step %d softmax output" % step, lp._get(g, "-1 n@"), lp._get(o, "-1 n@"), TOL)
            g.eval("bw\n"); o.eval("bw\n")
            gw = {n_: lp._get(g, e) for n_, e in G40}; go = {n_: lp._get(o, e) for n_, e in G40}
            do0, gdo0 = lp._get(o, "1 n@"), lp._get(g, "1 n@")       # dO of the conv layer = dX of the pool (in-place convention)
            flipped = lp.pool_flips("step %d dO conv1" % step, gdo0, do0, c1o)
            ties += len(flipped)
            ex_o = lp.conv_df64(img, do0); ex_g = lp.conv_df64(img, gdo0)
            for n_, _e in G40:
                if n_ in ("dw0", "db0"):
                    k_ = 0 if n_ == "dw0" else 1
                    check_tensor("step %d %s: oracle vs float64" % (step, n_), go[n_], ex_o[k_].reshape(go[n_].shape), TOL, floor=1e-2)
                    check_tensor("step %d %s: product vs float64 on its own operands" % (step, n_), gw[n_], ex_g[k_].reshape(gw[n_].shape), TOL)
                    if not flipped:
                        check_tensor("step %d %s" % (step, n_), gw[n_], go[n_], TOL, floor=1e-2)
                else:
                    check_tensor("step %d %s" % (step, n_), gw[n_], go[n_], TOL)
            keep = np.array([i not in flipped for i in range(img.shape[0])])
            check_tensor("step %d dX of layer 0 (produced on demand)" % step, lp._get(g, "0 n@")[keep], lp._get(o, "0 n@")[keep], TOL)
            g.eval("opt drop\n"); o.eval("opt drop\n")
            for n_, e in P40:
                po, pw = lp._get(o, e), lp._get(g, e)
                if flipped and n_ in ("w0", "b0"):
                    continue                                         # a verified tie: this step's filter update follows another (equally valid) winner
                _adam_check("step %d %s" % (step, n_), pw, po, 1e-3, g_first=go["d" + n_] if step == 0 else None, steps=step + 1)
                g.store(po, "net " + e); g.eval("drop drop")          # next step from ONE set of parameters (the oracle's)
        assert ties <= 4
    finally:
        ADAM_B1 = b1_old
        g.close(); o.close()


TB_40A = '''0 trace
256 28 28 1 nn.model 0.5 10 conv2d 2 maxpool relu flatten 100 linear relu 10 linear softmax constant md0
256 28 28 1 tensor rand constant img
: hot ( T -- T ) 256 0 do 1 i 10 * i 7 * 10 mod + t! loop ;
2560 vector zeros hot 256 1 10 1 reshape4 constant lbl
: histo ( M -- M ) 0 nn.w 30 s" nn/conv0" .histo 4 nn.w 30 s" nn/lin4" .histo 6 nn.w 30 s" nn/lin6" .histo ;
md0 img forward lbl backprop 0.001 nn.adam
1 .tbstep
img forward lbl loss.ce s" train/loss" .scalar
histo
img 16 s" mnist/train" .tile
drop
bye
'''


def test_t4_40a_tensorboard_words_read_live_hbm_tensors(tmp_path):
    """`ten4 -t <logdir>` on the GPU: the t4_40a script's `.scalar` / `.histo` / `.tile` words (examples/t4_40a.4th:16-31) after one Adam step of
    the N = 256 net - the histograms are built from weight tensors that live in HBM and were just updated by the fused launch plan.  Compared
    record by record with the file the oracle VM writes for the same script: same tags and steps, the image tile byte for byte (same Philox
    draw), the loss at 1e-4, histogram edges at 1e-4 of the range and bucket counts equal up to elements that sit on a bucket edge."""
    if not os.path.exists(TEN4_ORACLE):
        pytest.skip("oracle VM binary not shipped")
    import glob
    import subprocess
    from test_tensorboard_sink import records
    blobs = {}
    for name, binary in (("gpu", TEN4), ("cpu", TEN4_ORACLE)):
        d = tmp_path / name
        env = dict(os.environ, T4_SEED="40", T4_TB_FIXED_TIME="1700000000")
        r = subprocess.run([binary, "-t", str(d), "-r", "r"], input=TB_40A, capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0 and "check TensorBoard param" not in r.stdout, r.stdout[-2000:]
        blobs[name] = records(open(glob.glob(os.path.join(str(d), "r", "events.out.tfevents.*"))[0], "rb").read())
    gr, cr = blobs["gpu"], blobs["cpu"]
    assert len(gr) == len(cr) == 6                               # version, scalar, three histograms, tile
    assert gr[0] == cr[0] and gr[5] == cr[5]                     # header; the image tile of the batch (identical draw -> identical PNG bytes)

    import struct
    # scalar: last 4 bytes = the f32 value
    assert gr[1][:-4] == cr[1][:-4]
    lg, lc = struct.unpack("<f", gr[1][-4:])[0], struct.unpack("<f", cr[1][-4:])[0]
    assert abs(lg - lc) <= 1e-4 * abs(lc), (lg, lc)
    for i in (2, 3, 4):                                          # histograms: same length, every f64 field close, counts equal up to edge elements
        assert len(gr[i]) == len(cr[i])
        a = np.frombuffer(gr[i][-(31 * 8):], "<f8"); b = np.frombuffer(cr[i][-(31 * 8):], "<f8")      # the 31 bucket counts (packed doubles, last field)
        assert a.sum() == b.sum() and np.abs(a - b).sum() <= max(4.0, 2e-4 * a.sum()), (i, np.abs(a - b).sum())
