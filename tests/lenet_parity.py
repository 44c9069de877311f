"""Shared pieces of the LeNet (BASELINE configs #3 / #5) full-tensor parity tests: the net, float64 truth for the long conv-gradient
sums, verification of max-pool arg-max ties, and one training step of a product VM against the oracle VM."""
import numpy as np

from vm_util import check_tensor, rel_err

NET = "0.5 10 conv2d 2 maxpool relu 0.5 20 conv2d 0.5 dropout 2 maxpool relu flatten 100 linear 0.5 dropout 10 linear softmax"
PARAMS = [("w0", "0 nn.w"), ("b0", "0 nn.b"), ("w3", "3 nn.w"), ("b3", "3 nn.b"), ("w8", "8 nn.w"), ("b8", "8 nn.b"), ("w10", "10 nn.w"), ("b10", "10 nn.b")]
GRADS = [("dw0", "0 nn.dw"), ("db0", "0 nn.db"), ("dw3", "3 nn.dw"), ("db3", "3 nn.db"), ("dw8", "8 nn.dw"), ("db8", "8 nn.db"), ("dw10", "10 nn.dw"), ("db10", "10 nn.db")]
TOL = 1e-4                                                    # north_star: "outputs within 1e-4 relative of the reference"


def _setup(vm, n, row0, total):
    """the net at batch n, rows [row0, row0+n) of the whole batch's image draw, labels by GLOBAL row index"""
    out = vm.eval("0 trace\n%d 28 28 1 nn.model %s constant net\n" % (n, NET))
    off0 = vm.rand_tell()
    vm.rand_seek(off0 + row0 * 784)
    out += vm.eval("%d 28 28 1 tensor rand constant img\n" % n)
    vm.rand_seek(off0 + total * 784)
    out += vm.eval(": hot ( T -- T ) %d 0 do 1 i 10 * i %d + 7 * 10 mod + t! loop ;\n"
                   "%d vector zeros hot %d 1 10 1 reshape4 constant lbl\n"
                   ": fw ( N -- N ) img forward ;\n: bw ( N -- N ) lbl backprop ;\n: opt ( N -- N ) 0.01 0.0 nn.sgd ;\n" % (n, row0, n * 10, n))
    assert "?" not in out.replace("-> ok", ""), out
    return vm.rand_tell()


def _get(vm, expr):
    a = vm.fetch("net " + expr)                                # ( N -- N T )
    vm.eval("drop drop")
    return a


def conv_df64(X, dO, K=3, P=1):
    """dF / dB of a (K, 1, P) convolution in float64 from the reference's definition (nmath.tcu:211-338: dF is the un-flipped
    correlation of the layer input with dO, dB the sum of dO) - the exact value both fp32 implementations approximate"""
    X = np.asarray(X, np.float64); dO = np.asarray(dO, np.float64)
    N, H, W, C1 = X.shape
    Xp = np.zeros((N, H + 2 * P, W + 2 * P, C1)); Xp[:, P:P + H, P:P + W] = X
    dF = np.empty((C1, K, K, dO.shape[3]))
    for ky in range(K):
        for kx in range(K):
            dF[:, ky, kx, :] = np.tensordot(Xp[:, ky:ky + H, kx:kx + W, :], dO, axes=([0, 1, 2], [0, 1, 2]))
    return dF, dO.sum(axis=(0, 1, 2))


def pool_flips(name, got_dx, want_dx, fwd, tol=TOL):
    """dX of a 2x2 maxpool (= dO of the conv layer in front).  Max-pooling routes each gradient to the arg-max of its window; when
    the two largest forward values of a window agree to rounding (2 million windows per step here: it happens about once) the fp32
    summation order of the convolution decides which cell wins, in the reference as much as here.  Every element beyond `tol` must
    belong to such a tied window (top two values within 1e-5 relative, checked on the oracle's forward tensor `fwd`); returns the
    set of samples that had a flip."""
    got_dx = np.asarray(got_dx, np.float64); want_dx = np.asarray(want_dx, np.float64)
    bad = np.argwhere(np.abs(got_dx - want_dx) > tol * np.abs(want_dx).max())
    samples = set()
    for n, y, x, c in bad:
        win = np.sort(np.asarray(fwd[n, y // 2 * 2:y // 2 * 2 + 2, x // 2 * 2:x // 2 * 2 + 2, c], np.float64).ravel())
        assert abs(win[-1] - win[-2]) <= 1e-5 * max(abs(win[-1]), 1e-30), "%s: differs at %s away from an arg-max tie (window %s)" % (name, (n, y, x, c), win)
        samples.add(int(n))
    assert len(bad) <= 16, "%s: %d elements differ" % (name, len(bad))
    return samples


def _check_rows(name, got, want, skip, tol=TOL):
    """as _check, but the samples in `skip` (those with an arg-max flip upstream) are compared on their own and only loosely"""
    keep = np.array([i not in skip for i in range(want.shape[0])])
    _check(name, got[keep], want[keep], tol)


FLOOR = 1e-3                                                  # element-aware bar: |d| <= tol (|ref| + FLOOR max|ref|) on >= 99.99 % of the elements (vm_util.check_tensor)


def _check(name, got, want, tol=TOL, floor=None):
    check_tensor(name, got, want, tol, floor=max(floor or 0.0, FLOOR))   # max|d| / max|ref| <= tol AND the element-aware bar




def step_vs_oracle(g, o, img, step, lr=0.01):
    """ONE training step (words fw / bw / opt, see _setup) of product VM `g` against oracle VM `o`, both holding the same parameters
    and the image batch `img`: every gradient tensor, dX, the masks and the post-SGD parameters at TOL, with the arg-max-tie chain of
    test_gpu_config5_full (operands equal up to VERIFIED ties, each side within TOL of float64 on its own operands).  Leaves both VMs
    with the oracle's parameters (so the next call is again a one-step comparison).  Returns the set of samples that had a tie."""
    N = img.shape[0]
    g.eval("net fw\n"); o.eval("net fw\n")
    assert g.rand_tell() == o.rand_tell()
    c1o, c2o = _get(o, "1 n@"), _get(o, "5 n@")
    x3o, x3g = _get(o, "3 n@"), _get(g, "3 n@")
    _check("step %d conv2 input" % step, x3g, x3o)
    _check("step %d softmax output" % step, _get(g, "-1 n@"), _get(o, "-1 n@"))
    g.eval("bw\n"); o.eval("bw\n")
    for lab, e in (("mask_conv", "4 nn.ex"), ("mask_lin", "9 nn.ex")):
        assert np.array_equal(_get(g, e), _get(o, e)), lab
    gw = {n_: _get(g, e) for n_, e in GRADS}; go = {n_: _get(o, e) for n_, e in GRADS}
    do0, do3, gdo0, gdo3 = _get(o, "1 n@"), _get(o, "4 n@"), _get(g, "1 n@"), _get(g, "4 n@")
    flipped = pool_flips("step %d dX of pool 2" % step, _get(g, "5 n@"), _get(o, "5 n@"), c2o)
    _check_rows("step %d dO conv2" % step, gdo3, do3, flipped)
    ok = [i for i in range(N) if i not in flipped]
    flipped |= pool_flips("step %d dO conv1 = dX of pool 1" % step, gdo0[ok], do0[ok], c1o[ok])
    exact, exact_g = {}, {}
    exact["dw0"], exact["db0"] = conv_df64(img, do0); exact["dw3"], exact["db3"] = conv_df64(x3o, do3)
    exact_g["dw0"], exact_g["db0"] = conv_df64(img, gdo0); exact_g["dw3"], exact_g["db3"] = conv_df64(x3g, gdo3)
    for n_, e in GRADS:
        if n_ in exact:
            _check("step %d %s: oracle vs float64 on the oracle's operands" % (step, n_), go[n_], exact[n_].reshape(go[n_].shape), floor=1e-2)   # (the oracle sums ~1e5 terms sequentially in fp32)
            _check("step %d %s: product vs float64 on the product's operands" % (step, n_), gw[n_], exact_g[n_].reshape(gw[n_].shape))
            if not flipped:
                _check("step %d %s: product vs oracle (no arg-max tie in this step)" % (step, n_), gw[n_], go[n_], floor=1e-2)
        else:
            _check("step %d %s" % (step, n_), gw[n_], go[n_])
    _check_rows("step %d dx" % step, _get(g, "0 n@"), _get(o, "0 n@"), flipped)
    before = {n_: _get(o, e) for n_, e in PARAMS}
    g.eval("opt drop\n"); o.eval("opt drop\n")
    for n_, e in PARAMS:
        po, pw = _get(o, e), _get(g, e)
        nw = po.shape[0]                                     # k_sgd divides by the parameter tensor's N(): C1 for a conv filter, else 1 (quirk a-19, gradient.cu:135-137)
        if "d" + n_ in exact and flipped:
            _check("step %d %s: product vs w - lr dw(float64)" % (step, n_), pw, before[n_] - lr * exact_g["d" + n_].reshape(pw.shape) / nw)
        else:
            _check("step %d %s" % (step, n_), pw, po)
        g.store(po, "net " + e); g.eval("drop drop")
    return flipped
