"""GPU tests of the work a conv-stack backward may DEFER (round 4): the dF | dB partial fold inside the optimizer launch (t4k_opt_step) and the
lazily produced dX of the first conv layer (t4k_conv_stack_dx0).  The bar: whatever the caller does next, it sees what the undeferred
path leaves - bit for bit for the fold (same arithmetic, same order), 1e-4 against the oracle for the lazily produced dX."""
import ctypes

import numpy as np
import pytest

from test_gpu_conv_stack import CASES, _build, _oracle_backward, _oracle_forward, _params
from test_gpu_parity import Dev, p, rel

pytestmark = pytest.mark.gpu


class ParamRec(ctypes.Structure):
    _fields_ = [("G", ctypes.c_void_p), ("DG", ctypes.c_void_p), ("M", ctypes.c_void_p), ("V", ctypes.c_void_p),
                ("n", ctypes.c_long), ("Nw", ctypes.c_int), ("pad", ctypes.c_int)]


@pytest.fixture(scope="module")
def dev(t4k):
    return Dev(t4k)


def _launches(t4k):
    t4k.lib.t4k_launch_count.restype = ctypes.c_ulonglong
    return t4k.lib.t4k_launch_count()


def _setup(t4k, dev, oracle, case, seed):
    N, H, W, Cin, stages, flat = CASES[case]
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    params = _params(rng, Cin, stages)
    ref, _end = _oracle_forward(oracle, X, stages, flat, params, 55 + case, 4096)
    arr, bufs = _build(dev, oracle, X, stages, flat, params, ref)
    assert t4k.lib.t4k_conv_stack_ok(arr, len(stages), N) == 1
    DY = rng.standard_normal(ref[-1]["last"].shape).astype(np.float32)
    return N, X, stages, flat, params, arr, bufs, dev.up(DY), DY


def _table(dev, bufs, params, extra=None):
    """device + host parameter table over every F / B of the stack (+ an extra plain tensor, like a linear layer's), momentum tensors"""
    recs = []; chunks = 0; mom = []
    items = []
    for si, (F, B) in enumerate(params):
        items += [(bufs[si]["F"], bufs[si]["DF"], F.size, F.shape[0]), (bufs[si]["B"], bufs[si]["DB"], B.size, B.size)]
    if extra is not None:
        items.append(extra)
    for G, DG, n, nw in items:
        m, v = dev.zeros(n), dev.zeros(n)
        mom += [m, v]
        recs.append(ParamRec(p(G), p(DG), p(m), p(v), n, nw, chunks)); chunks += (n + 1023) // 1024
    host = (ParamRec * len(recs))(*recs)
    raw = np.frombuffer(bytes(host), np.uint8).copy()
    return host, dev.up(raw), len(recs), chunks, mom


@pytest.mark.parametrize("kind,b1", [(0, 0.0), (0, 0.9), (1, 0.9), (2, 0.9)], ids=["sgd", "sgd-momentum", "adam", "adamw"])
@pytest.mark.parametrize("case", [0, 2, 3])
def test_partial_fold_inside_the_optimizer_launch_is_bit_identical(t4k, dev, oracle, case, kind, b1):
    """t4k_conv_stack_bwd(train | 4) + t4k_opt_step (2 launches) against t4k_conv_stack_bwd(train) + t4k_opt_chunked (3 launches): parameters,
    momenta and the zeroed gradients agree bit for bit; a tensor that is NOT part of the fold (a linear layer's) is updated as before."""
    res = []
    for defer in (False, True):
        N, X, stages, flat, params, arr, bufs, dDY, _ = _setup(t4k, dev, oracle, case, 500 + case)
        rng = np.random.default_rng(9)
        w_lin = (rng.standard_normal(3000)).astype(np.float32); g_lin = rng.standard_normal(3000).astype(np.float32)
        dW, dG = dev.up(w_lin), dev.up(g_lin)
        host, tab, nt, chunks, mom = _table(dev, bufs, params, extra=(dW, dG, 3000, 1))
        for si in range(len(stages)):
            bufs[si]["DF"].fill_(0.25); bufs[si]["DB"].fill_(-0.5)              # gradients ACCUMULATE: the fold adds to what is there
        t4k.call("t4k_rand_init", 55 + case); t4k.call("t4k_rand_set_offset", 4096)
        bufs[0]["X"].copy_(dev.torch.from_numpy(X))
        t4k.call("t4k_conv_stack_fwd", p(bufs[0]["X"]), None, arr, len(stages), N, None)
        t4k.call("t4k_sync", None)
        l0 = _launches(t4k)
        t4k.call("t4k_conv_stack_bwd", p(dDY), arr, len(stages), N, 5 if defer else 1, None)
        for step in range(2):                                                    # a second optimizer call finds nothing pending: plain update of zeroed gradients
            if defer:
                t4k.call("t4k_opt_step", kind, p(tab), host, nt, chunks, ctypes.c_float(0.01), ctypes.c_float(b1), ctypes.c_float(0.999), ctypes.c_float(0.01), None)
            else:
                t4k.call("t4k_opt_chunked", kind, p(tab), nt, chunks, ctypes.c_float(0.01), ctypes.c_float(b1), ctypes.c_float(0.999), ctypes.c_float(0.01), None)
            if step == 0:
                assert _launches(t4k) - l0 == (2 if defer else 3)
        out = {}
        for si in range(len(stages)):
            for k_ in ("F", "B", "DF", "DB"):
                out["%d%s" % (si, k_)] = dev.down(bufs[si][k_]).copy()
        out["lin"] = dev.down(dW).copy(); out["lin_g"] = dev.down(dG).copy()
        for i, m in enumerate(mom):
            out["m%d" % i] = dev.down(m).copy()
        res.append(out)
    for k_ in res[0]:
        assert np.array_equal(res[0][k_], res[1][k_]), k_
    assert not np.array_equal(res[0]["0F"], params[0][0])                        # ... and the step did move the parameters
    assert float(np.abs(res[1]["0DF"]).max()) == 0.0 and float(np.abs(res[1]["lin_g"]).max()) == 0.0


@pytest.mark.parametrize("case", [0, 2])
def test_any_other_entry_point_runs_the_deferred_fold_first(t4k, dev, oracle, case):
    """a caller that reads a gradient tensor (any t4k_* call: here the sync in front of the read-back, a reduction, a copy) between the deferred backward
    and the optimizer sees the folded gradients - the same bits the undeferred backward leaves"""
    got = []
    for defer in (False, True):
        N, X, stages, flat, params, arr, bufs, dDY, _ = _setup(t4k, dev, oracle, case, 600 + case)
        for si in range(len(stages)):
            bufs[si]["DF"].fill_(0.25); bufs[si]["DB"].fill_(-0.5)
        t4k.call("t4k_rand_init", 55 + case); t4k.call("t4k_rand_set_offset", 4096)
        bufs[0]["X"].copy_(dev.torch.from_numpy(X))
        t4k.call("t4k_conv_stack_fwd", p(bufs[0]["X"]), None, arr, len(stages), N, None)
        t4k.call("t4k_conv_stack_bwd", p(dDY), arr, len(stages), N, 5 if defer else 1, None)
        cp = dev.zeros(params[-1][0].shape)
        t4k.call("t4k_copy", p(bufs[-1]["DF"]), p(cp), params[-1][0].size, None)   # an unrelated-looking entry point: the fold runs in front of it
        got.append([dev.down(cp).copy()] + [dev.down(bufs[si][k_]).copy() for si in range(len(stages)) for k_ in ("DF", "DB")])
    for a, b in zip(*got):
        assert np.array_equal(a, b)
    assert rel(got[1][0], got[1][-2]) == 0.0


@pytest.mark.parametrize("case", [0, 2, 4])
def test_lazy_dx_of_the_first_layer_is_produced_on_demand(t4k, dev, oracle, case):
    """t4k_conv_stack_bwd(train | 8) skips the first layer's dX; t4k_conv_stack_dx0 produces it later - even after the optimizer has changed
    the filter - and X / DXS then hold what the eager backward stores (oracle: 1e-4; the eager GPU path: a few ulp, another kernel).  Every
    other tensor of the backward is untouched by the flag, and the next forward ends the offer."""
    N, X, stages, flat, params, arr, bufs, dDY, DY = _setup(t4k, dev, oracle, case, 700 + case)

    def fwd():
        t4k.call("t4k_rand_init", 55 + case); t4k.call("t4k_rand_set_offset", 4096)
        bufs[0]["X"].copy_(dev.torch.from_numpy(X))
        t4k.call("t4k_conv_stack_fwd", p(bufs[0]["X"]), None, arr, len(stages), N, None)
    fwd()
    got_fwd = []; x = X
    for si, st_ in enumerate(stages):
        t = {k_: dev.down(bufs[si][k_]).reshape(v.shape).copy() for k_, v in _oracle_forward(oracle, X, stages, flat, params, 55 + case, 4096)[0][si].items() if k_ in bufs[si]}
        t["in"] = x
        C0, K, pre, pool, post = st_
        x = t["post_out"] if post else (t["pool_out"] if pool else (t["pre_out"] if pre else t["O"]))
        t["last"] = x; got_fwd.append(t)
    want = _oracle_backward(oracle, got_fwd, stages, flat, params, DY)
    # eager
    t4k.call("t4k_conv_stack_bwd", p(dDY), arr, len(stages), N, 1, None)
    eager = {"X": dev.down(bufs[0]["X"]).copy(), "DF0": dev.down(bufs[0]["DF"]).copy(), "O0": dev.down(bufs[0]["O"]).copy()}
    assert t4k.lib.t4k_conv_stack_dx0_pending(ctypes.c_void_p(p(bufs[0]["O"]))) == 0
    # lazy
    for si in range(len(stages)):
        bufs[si]["DF"].zero_(); bufs[si]["DB"].zero_()
    fwd()
    bufs[0]["DXS"].fill_(7.0)
    t4k.call("t4k_conv_stack_bwd", p(dDY), arr, len(stages), N, 1 | 8, None)
    banded = t4k.lib.t4k_conv_stack_dx0_pending(ctypes.c_void_p(p(bufs[0]["O"]))) == 1
    if not banded:                                                   # a geometry without a banded backward computes dX eagerly whatever the flag says
        assert np.array_equal(dev.down(bufs[0]["X"]), eager["X"])
        return
    assert np.array_equal(dev.down(bufs[0]["DF"]), eager["DF0"]) and np.array_equal(dev.down(bufs[0]["O"]), eager["O0"])
    assert float(dev.down(bufs[0]["DXS"]).min()) == 7.0              # skipped: nothing stored
    F0 = dev.down(bufs[0]["F"]).copy()
    bufs[0]["F"].mul_(1.5)                                           # the optimizer moves the filter before anybody asks for dX ...
    t4k.call("t4k_conv_stack_dx0", arr, N, None)
    assert t4k.lib.t4k_conv_stack_dx0_pending(ctypes.c_void_p(p(bufs[0]["O"]))) == 0
    for k_ in ("X", "DXS"):
        g = dev.down(bufs[0][k_])
        assert rel(g, want[0]["DX"]) < 1e-4, (k_, rel(g, want[0]["DX"]))           # ... and it is still the dX of the filter that backward used
        assert rel(g, eager["X"]) < 2e-6, (k_, rel(g, eager["X"]))
    bufs[0]["F"].copy_(dev.torch.from_numpy(F0))
    # the next forward ends the offer
    fwd()
    t4k.call("t4k_conv_stack_bwd", p(dDY), arr, len(stages), N, 1 | 8, None)
    assert t4k.lib.t4k_conv_stack_dx0_pending(ctypes.c_void_p(p(bufs[0]["O"]))) == 1
    fwd()
    assert t4k.lib.t4k_conv_stack_dx0_pending(ctypes.c_void_p(p(bufs[0]["O"]))) == 0
    before = dev.down(bufs[0]["DXS"]).copy()
    t4k.call("t4k_conv_stack_dx0", arr, N, None)                     # nothing pending: a no-op
    assert np.array_equal(dev.down(bufs[0]["DXS"]), before)
