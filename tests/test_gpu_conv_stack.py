"""GPU parity of the sample-resident convolution stack (csrc/conv_stack.hip: t4k_conv_stack_fwd / _bwd) against the oracle's
SEPARATE layers - conv2d, activate / dropout (Philox slice of t4k_rand), pool, dpool, mask multiplies, conv2d backward - on every
tensor the separate layers write.  Forward: 1e-4 relative (dropout masks and the layer-0 copy bit-exact, Philox stream advanced
identically).  Backward: the GPU buffers are loaded with the ORACLE's forward state first, so arg-max positions and derivative masks are
the same on both sides and every dX / dF / dB is compared at 1e-4 relative."""
import ctypes

import numpy as np
import pytest

from test_gpu_parity import Dev, PoolBlock, p, rel

pytestmark = pytest.mark.gpu
RTOL = 1e-4


class ConvStage(ctypes.Structure):
    _fields_ = [("F", ctypes.c_void_p), ("B", ctypes.c_void_p), ("O", ctypes.c_void_p), ("DF", ctypes.c_void_p), ("DB", ctypes.c_void_p),
                ("X", ctypes.c_void_p), ("DXS", ctypes.c_void_p),
                ("H", ctypes.c_int), ("W", ctypes.c_int), ("C1", ctypes.c_int), ("C0", ctypes.c_int), ("K", ctypes.c_int),
                ("run", PoolBlock)]


@pytest.fixture(scope="module")
def dev(t4k):
    return Dev(t4k)


def _lay(oracle):
    return {"dropout": (oracle.L_DROPOUT, 0.5), "relu": (oracle.L_RELU, 0.0), "leaky": (oracle.L_LEAKYRL, 0.1), "tanh": (oracle.L_TANH, 0.0),
            "elu": (oracle.L_ELU, 1.0), "max": oracle.L_MAXPOOL, "avg": oracle.L_AVGPOOL, "min": oracle.L_MINPOOL}


# (N, H, W, C_in, [(C0, K, pre, pool, post)], flatten)
CASES = [
    (128, 28, 28, 1, [(10, 3, None, "max", "relu"), (20, 3, "dropout", "max", "relu")], True),      # the LeNet front end of bench.py
    (5, 12, 8, 3, [(8, 5, "relu", "avg", None), (6, 3, None, None, "tanh")], False),               # 5x5, non-square, a stage without pool
    (3, 16, 16, 2, [(7, 3, "leaky", "max", "dropout"), (9, 3, None, "min", "elu"), (4, 3, "dropout", None, None)], True),   # 3 stages, odd channels
    (2, 8, 8, 4, [(32, 3, None, "max", "relu")], True),                                             # one stage, two channel tiles
    (1, 6, 6, 1, [(3, 3, None, None, None), (5, 3, None, "max", None)], False),                     # a bare conv in front
]


def _oracle_forward(oracle, X, stages, flat, params, seed, off):
    """the separate layers; returns per stage a dict of every tensor written"""
    o = oracle.lib(); P = oracle.P; LAY = _lay(oracle)
    o.t4o_rand_init(seed); o.t4o_rand_set_offset(off)
    N = X.shape[0]
    x = X; out = []
    for si, (C0, K, pre, pool, post) in enumerate(stages):
        F, B = params[si]
        H, W, C1 = x.shape[1:]
        Y = np.zeros((N, H, W, C0), np.float32)
        assert o.t4o_conv2d_fwd(P(x), P(Y), P(F), P(B), N, H, W, C1, H, W, C0, K, 1, K // 2) == 0
        t = {"in": x, "O": Y}; cur = Y
        if pre:
            L, a = LAY[pre]; f = np.zeros(cur.size, np.float32); y = np.zeros_like(cur)
            if pre == "dropout":
                o.t4o_rand(P(f), f.size, 0, 0.0, 1.0)
            o.t4o_activate(L, P(cur), P(y), P(f), a, cur.size); t["pre_mask"] = f.reshape(cur.shape); t["pre_out"] = y; cur = y
        if pool:
            q = np.zeros((N, H // 2, W // 2, C0), np.float32)
            o.t4o_pool(LAY[pool], P(cur), P(q), N, H, W, H // 2, W // 2, C0, 2); t["pool_out"] = q; cur = q
        if post:
            L, a = LAY[post]; f = np.zeros(cur.size, np.float32); y = np.zeros_like(cur)
            if post == "dropout":
                o.t4o_rand(P(f), f.size, 0, 0.0, 1.0)
            o.t4o_activate(L, P(cur), P(y), P(f), a, cur.size); t["post_mask"] = f.reshape(cur.shape); t["post_out"] = y; cur = y
        if flat and si == len(stages) - 1:
            t["copy_out"] = cur.copy()
        t["last"] = cur
        out.append(t); x = cur
    return out, o.t4o_rand_offset()


def _build(dev, oracle, X, stages, flat, params, ref):
    """device buffers + the t4k_conv_stage array"""
    LAY = _lay(oracle)
    arr = (ConvStage * len(stages))()
    bufs = []
    for si, (C0, K, pre, pool, post) in enumerate(stages):
        t = ref[si]; d = {k: dev.zeros(v.shape) for k, v in t.items() if k not in ("in", "last")}
        F, B = params[si]
        d["F"], d["B"] = dev.up(F), dev.up(B)
        d["DF"], d["DB"] = dev.zeros(F.shape), dev.zeros(B.shape)
        # the conv's input tensor: for s > 0 it IS the last tensor of the run in front (the model's layers share it)
        prev_last = None
        if si > 0:
            pt = stages[si - 1]; pd = bufs[si - 1]
            prev_last = pd["post_out"] if pt[4] else (pd["pool_out"] if pt[3] else (pd["pre_out"] if pt[2] else pd["O"]))
        d["X"] = prev_last if prev_last is not None else dev.zeros(t["in"].shape); d["DXS"] = dev.zeros(t["in"].shape)
        s = arr[si]
        s.F, s.B, s.O, s.DF, s.DB, s.X, s.DXS = p(d["F"]), p(d["B"]), p(d["O"]), p(d["DF"]), p(d["DB"]), p(d["X"]), p(d["DXS"])
        s.H, s.W, s.C1, s.C0, s.K = t["in"].shape[1], t["in"].shape[2], t["in"].shape[3], C0, K
        b = s.run; b.KS = 2 if pool else 1
        if pre:
            b.pre_layer, b.pre_alpha = LAY[pre]; b.pre_mask = p(d["pre_mask"]); b.pre_out = p(d["pre_out"])
        if pool:
            b.pool_layer = LAY[pool]; b.pool_out = p(d["pool_out"])
        if post:
            b.post_layer, b.post_alpha = LAY[post]; b.post_mask = p(d["post_mask"]); b.post_out = p(d["post_out"])
        if "copy_out" in t:
            b.copy_out = p(d["copy_out"])
        bufs.append(d)
    return arr, bufs


def _params(rng, Cin, stages):
    out = []; c1 = Cin
    for C0, K, *_ in stages:
        out.append(((rng.standard_normal((c1, K, K, C0)) * 0.3).astype(np.float32), rng.standard_normal(C0).astype(np.float32)))
        c1 = C0
    return out


@pytest.mark.parametrize("case", range(len(CASES)))
def test_conv_stack_forward_matches_the_separate_layers(t4k, dev, oracle, case):
    N, H, W, Cin, stages, flat = CASES[case]
    rng = np.random.default_rng(100 + case)
    X = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    params = _params(rng, Cin, stages)
    seed, off = 1234 + case, 4096 * (case + 1)
    ref, end = _oracle_forward(oracle, X, stages, flat, params, seed, off)
    arr, bufs = _build(dev, oracle, X, stages, flat, params, ref)
    assert t4k.lib.t4k_conv_stack_ok(arr, len(stages), N) == 1
    t4k.call("t4k_rand_init", seed); t4k.call("t4k_rand_set_offset", off)
    dX, dX0 = dev.up(X), dev.zeros(X.shape)
    t4k.call("t4k_conv_stack_fwd", p(dX), p(dX0), arr, len(stages), N, None)
    assert np.array_equal(dev.down(dX0), X)                                  # the model's layer-0 copy
    assert t4k.lib.t4k_rand_offset() == end                                  # Philox stream advanced exactly as the separate layers do
    for si, (C0, K, pre, pool, post) in enumerate(stages):
        t, d = ref[si], bufs[si]
        for k_ in ("O", "pre_mask", "pre_out", "pool_out", "post_mask", "post_out", "copy_out"):
            if k_ not in t:
                continue
            got = dev.down(d[k_]).reshape(t[k_].shape)
            if k_.endswith("mask"):
                which = pre if k_ == "pre_mask" else post
                if which == "dropout":
                    assert np.array_equal(got, t[k_]), "stage %d %s" % (si, k_)   # same Philox slice: bit-exact
                else:
                    # derivative masks flip where a pre-activation sits within rounding of the kink: all but a few elements equal
                    assert np.mean(np.abs(got - t[k_]) > 1e-3) < 1e-4, "stage %d %s" % (si, k_)
                continue
            assert rel(got, t[k_]) < RTOL, "stage %d %s: %.3g" % (si, k_, rel(got, t[k_]))


def _oracle_backward(oracle, ref, stages, flat, params, DY):
    """the separate layers in reverse, in the reference's in-place convention; returns per stage the buffers after backprop"""
    o = oracle.lib(); P = oracle.P; LAY = _lay(oracle)
    out = [None] * len(stages)
    g = DY
    for si in range(len(stages) - 1, -1, -1):
        C0, K, pre, pool, post = stages[si]
        t = {k: v.copy() for k, v in ref[si].items()}
        F, B = params[si]
        X = t["in"]; N, H, W, C1 = X.shape
        last = "post_out" if post else ("pool_out" if pool else ("pre_out" if pre else "O"))
        g = g.reshape(t[last].shape)
        if "copy_out" in t:
            t[last][...] = g                                                  # flatten: in = out
        if post:
            tgt = "pool_out" if pool else ("pre_out" if pre else "O")
            r = np.zeros(g.size, np.float32); o.t4o_tt_op(oracle.MUL, P(np.ascontiguousarray(g)), P(t["post_mask"]), P(r), g.size)
            t[tgt][...] = r.reshape(t[tgt].shape); g = t[tgt].copy()
        if pool:
            tgt = "pre_out" if pre else "O"
            o.t4o_dpool(LAY[pool], P(t[tgt]), P(np.ascontiguousarray(g)), N, H, W, H // 2, W // 2, C0, 2); g = t[tgt].copy()
        if pre:
            r = np.zeros(g.size, np.float32); o.t4o_tt_op(oracle.MUL, P(np.ascontiguousarray(g)), P(t["pre_mask"]), P(r), g.size)
            t["O"][...] = r.reshape(t["O"].shape); g = t["O"].copy()
        DX = np.zeros_like(X); DF = np.zeros_like(F); DB = np.zeros_like(B)
        assert o.t4o_conv2d_bwd(P(X), P(np.ascontiguousarray(g)), P(DX), P(F), P(DF), P(DB), N, H, W, C1, H, W, C0, K, 1, K // 2, 1) == 0
        t["DX"], t["DF"], t["DB"] = DX, DF, DB
        out[si] = t; g = DX
    return out


@pytest.mark.parametrize("stale", [False, True], ids=["cold", "stale-saved-state"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_conv_stack_backward_matches_the_separate_layers(t4k, dev, oracle, case, stale):
    """backward from the layer tensors alone (whole-image kernel).  `stale`: a stack forward of ANOTHER batch ran first and left its
    saved conv inputs / arg-max codes behind, then the layer tensors were rewritten by some other path (here: loaded from the oracle);
    the caller says so with train | 2 and the backward must not touch the saved state."""
    N, H, W, Cin, stages, flat = CASES[case]
    if any(a in ("sigmoid",) for st_ in stages for a in st_[2:]):
        pytest.skip("pass-through activation")
    rng = np.random.default_rng(200 + case)
    X = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    params = _params(rng, Cin, stages)
    ref, _end = _oracle_forward(oracle, X, stages, flat, params, 77 + case, 8192)
    arr, bufs = _build(dev, oracle, X, stages, flat, params, ref)
    assert t4k.lib.t4k_conv_stack_ok(arr, len(stages), N) == 1
    if stale:
        other = dev.up(rng.standard_normal((N, H, W, Cin)).astype(np.float32) * 3)
        t4k.call("t4k_conv_stack_fwd", p(other), p(bufs[0]["X"]), arr, len(stages), N, None)
    # load the ORACLE's forward state: same arg-max positions and masks on both sides
    for si in range(len(stages)):
        for k_, v in ref[si].items():
            if k_ in bufs[si]:
                bufs[si][k_].copy_(dev.torch.from_numpy(np.ascontiguousarray(v)))
        if si == 0:
            bufs[0]["X"].copy_(dev.torch.from_numpy(X))
        # pre-existing gradient values: the fold must ADD (gradient accumulation over several backprops)
        bufs[si]["DF"].fill_(0.25); bufs[si]["DB"].fill_(-0.5)
    DY = rng.standard_normal(ref[-1]["last"].shape).astype(np.float32)
    want = _oracle_backward(oracle, ref, stages, flat, params, DY)
    t4k.call("t4k_conv_stack_bwd", p(dev.up(DY)), arr, len(stages), N, 3 if stale else 1, None)
    for si in range(len(stages) - 1, -1, -1):
        t, d = want[si], bufs[si]
        C0, K, pre, pool, post = stages[si]
        assert rel(dev.down(d["DXS"]), t["DX"]) < RTOL, "stage %d dX (scratch copy): %.3g" % (si, rel(dev.down(d["DXS"]), t["DX"]))
        assert np.array_equal(dev.down(d["X"]), dev.down(d["DXS"])), "stage %d: in = dx" % si
        assert rel(dev.down(d["DF"]) - 0.25, t["DF"]) < RTOL, "stage %d dF: %.3g" % (si, rel(dev.down(d["DF"]) - 0.25, t["DF"]))
        assert rel(dev.down(d["DB"]) + 0.5, t["DB"]) < RTOL, "stage %d dB: %.3g" % (si, rel(dev.down(d["DB"]) + 0.5, t["DB"]))
        last = "post_out" if post else ("pool_out" if pool else ("pre_out" if pre else "O"))
        for k_ in ("O", "pre_out", "pool_out", "post_out"):
            if k_ not in t:
                continue
            if si + 1 < len(stages) and k_ == last:
                continue                                   # shared with the next stage's input tensor: holds that stage's dX (checked above)
            if k_ == last and "copy_out" not in t:
                continue                                   # the run's last tensor without a flatten behind it: nothing writes it
            got = dev.down(d[k_]).reshape(t[k_].shape)
            assert rel(got, t[k_]) < RTOL, "stage %d bwd %s: %.3g" % (si, k_, rel(got, t[k_]))


@pytest.mark.parametrize("case", range(len(CASES)))
def test_conv_stack_forward_then_backward_uses_what_the_forward_saved(t4k, dev, oracle, case):
    """forward THEN backward on the GPU: the backward now runs banded (several workgroups per image) on what the forward saved - a copy
    of every conv input and the pool arg-max codes - instead of reading layer tensors a neighbouring band overwrites in place.  The
    oracle's separate-layer backward is fed the GPU's own forward tensors, so masks and arg-max positions agree by construction and
    every dX / dF / dB must match at 1e-4."""
    N, H, W, Cin, stages, flat = CASES[case]
    rng = np.random.default_rng(300 + case)
    X = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    params = _params(rng, Cin, stages)
    ref, _end = _oracle_forward(oracle, X, stages, flat, params, 55 + case, 4096)
    arr, bufs = _build(dev, oracle, X, stages, flat, params, ref)
    assert t4k.lib.t4k_conv_stack_ok(arr, len(stages), N) == 1
    t4k.call("t4k_rand_init", 55 + case); t4k.call("t4k_rand_set_offset", 4096)
    bufs[0]["X"].copy_(dev.torch.from_numpy(X))
    t4k.call("t4k_conv_stack_fwd", p(bufs[0]["X"]), None, arr, len(stages), N, None)       # (the batch already sits in the layer-0 tensor)
    got_fwd = []
    x = X
    for si, st_ in enumerate(stages):
        t = {k_: dev.down(bufs[si][k_]).reshape(v.shape).copy() for k_, v in ref[si].items() if k_ in bufs[si]}
        t["in"] = x
        C0, K, pre, pool, post = st_
        x = t["post_out"] if post else (t["pool_out"] if pool else (t["pre_out"] if pre else t["O"]))
        t["last"] = x
        got_fwd.append(t)
    for si in range(len(stages)):
        bufs[si]["DF"].fill_(0.25); bufs[si]["DB"].fill_(-0.5)
    DY = rng.standard_normal(got_fwd[-1]["last"].shape).astype(np.float32)
    want = _oracle_backward(oracle, got_fwd, stages, flat, params, DY)
    t4k.call("t4k_conv_stack_bwd", p(dev.up(DY)), arr, len(stages), N, 1, None)
    for si in range(len(stages) - 1, -1, -1):
        t, d = want[si], bufs[si]
        C0, K, pre, pool, post = stages[si]
        assert rel(dev.down(d["DXS"]), t["DX"]) < RTOL, "stage %d dX: %.3g" % (si, rel(dev.down(d["DXS"]), t["DX"]))
        assert np.array_equal(dev.down(d["X"]), dev.down(d["DXS"])), "stage %d: in = dx" % si
        assert rel(dev.down(d["DF"]) - 0.25, t["DF"]) < RTOL, "stage %d dF: %.3g" % (si, rel(dev.down(d["DF"]) - 0.25, t["DF"]))
        assert rel(dev.down(d["DB"]) + 0.5, t["DB"]) < RTOL, "stage %d dB: %.3g" % (si, rel(dev.down(d["DB"]) + 0.5, t["DB"]))
        last = "post_out" if post else ("pool_out" if pool else ("pre_out" if pre else "O"))
        for k_ in ("O", "pre_out", "pool_out", "post_out"):
            if k_ not in t or (si + 1 < len(stages) and k_ == last) or (k_ == last and "copy_out" not in t):
                continue
            assert rel(dev.down(d[k_]).reshape(t[k_].shape), t[k_]) < RTOL, "stage %d bwd %s: %.3g" % (si, k_, rel(dev.down(d[k_]).reshape(t[k_].shape), t[k_]))


def test_conv_stack_refuses_what_it_cannot_hold(t4k, dev, oracle):
    arr = (ConvStage * 1)()
    s = arr[0]; s.H, s.W, s.C1, s.C0, s.K = 32, 32, 64, 64, 3; s.run.KS = 1
    s.F = s.B = s.O = 1                                                      # non-null placeholders: the check is on shapes
    assert t4k.lib.t4k_conv_stack_ok(arr, 1, 4) == 0                         # 64 channels: the LDS-staged MFMA GEMM kernels' territory
    s.C1, s.C0, s.K = 4, 4, 4
    assert t4k.lib.t4k_conv_stack_ok(arr, 1, 4) == 0                         # even kernel size


def test_backprop_after_a_traced_forward_ignores_what_an_earlier_stack_forward_saved():
    """host side of the `train | 2` flag (model.cpp run_forward / run_backward): a fused forward of batch A leaves its conv inputs and
    arg-max codes saved; a forward of batch B under `1 trace` takes the separate layer kernels; the backprop that follows is fused
    again and must differentiate B's forward, not A's.  Compared with a VM that ran B's forward through the stack kernel (same
    Philox positions, so the same dropout masks)."""
    from lenet_parity import GRADS, NET, TOL, _get
    from tensorforth_amd.vm import VM
    from vm_util import rel_err
    N = 32
    got = {}
    for traced in (False, True):
        vm = VM(device=0, seed=777)
        try:
            out = vm.eval("0 trace\n%d 28 28 1 nn.model %s constant net\n%d 28 28 1 tensor rand constant imgA\n%d 28 28 1 tensor rand constant imgB\n"
                          ": hot ( T -- T ) %d 0 do 1 i 10 * i 7 * 10 mod + t! loop ;\n%d vector zeros hot %d 1 10 1 reshape4 constant lbl\n"
                          % (N, NET, N, N, N, N * 10, N))
            assert "?" not in out.replace("-> ok", ""), out
            vm.eval("net imgA forward drop\n")                                     # fused: saves A's state
            pos = vm.rand_tell()
            vm.eval("1 trace net imgB forward drop 0 trace\n" if traced else "net imgB forward drop\n")
            assert vm.rand_tell() > pos
            vm.eval("net lbl backprop drop\n")
            got[traced] = {n_: _get(vm, e) for n_, e in GRADS}
            got[traced]["dx"] = _get(vm, "0 n@")
        finally:
            vm.close()
    for n_ in got[False]:
        e = rel_err(got[True][n_], got[False][n_])
        assert e <= TOL, "%s after a traced forward: %.3g" % (n_, e)


class StackHead(ctypes.Structure):
    _fields_ = [("W1", ctypes.c_void_p), ("B1", ctypes.c_void_p), ("Y1", ctypes.c_void_p), ("mid_layer", ctypes.c_int), ("mid_alpha", ctypes.c_float),
                ("mid_mask", ctypes.c_void_p), ("mid_out", ctypes.c_void_p), ("W2", ctypes.c_void_p), ("B2", ctypes.c_void_p), ("Y2", ctypes.c_void_p),
                ("P", ctypes.c_void_p), ("E1", ctypes.c_int), ("E0a", ctypes.c_int), ("E0b", ctypes.c_int),
                ("label", ctypes.c_void_p), ("hot", ctypes.c_void_p), ("hit_flag", ctypes.c_void_p), ("n_label", ctypes.c_int)]


@pytest.mark.parametrize("case,EA,EB,mid", [(0, 100, 10, "dropout"), (0, 100, 10, "relu"), (2, 37, 5, "dropout"), (3, 64, 16, None), (2, 130, 3, "tanh")])
def test_conv_stack_with_classifier_head_in_one_launch(t4k, dev, oracle, case, EA, EB, mid):
    """t4k_conv_stack_head_fwd == the oracle's separate layers: the stack (every tensor, as above), then linear E1 -> EA, the element-wise
    layer (dropout mask bit-exact: same Philox slice behind the stack's own draws), linear EA -> EB, softmax - at 1e-4 relative; run twice
    (the accumulators / tickets the bands share must be left clean) and the Philox stream ends where the separate layers' ends."""
    N, H, W, Cin, stages, flat = CASES[case]
    assert flat
    o = oracle.lib(); P = oracle.P; LAY = _lay(oracle)
    rng = np.random.default_rng(500 + case + EA)
    X = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    params = _params(rng, Cin, stages)
    seed, off = 4321 + case, 8192
    ref, _ = _oracle_forward(oracle, X, stages, flat, params, seed, off)
    xf = ref[-1]["last"].reshape(N, -1); E1 = xf.shape[1]
    W1 = (rng.standard_normal((EA, E1)) * 0.1).astype(np.float32); B1 = rng.standard_normal(EA).astype(np.float32)
    W2 = (rng.standard_normal((EB, EA)) * 0.3).astype(np.float32); B2 = rng.standard_normal(EB).astype(np.float32)
    Y1 = np.zeros((N, EA), np.float32); assert o.t4o_linear_fwd(P(np.ascontiguousarray(xf)), P(W1), P(B1), P(Y1), N, EA, E1) == 0
    cur = Y1; Fm = Am = None
    if mid:
        L, a = LAY[mid]; Fm = np.zeros(Y1.size, np.float32); Am = np.zeros_like(Y1)
        if mid == "dropout":
            o.t4o_rand(P(Fm), Fm.size, 0, 0.0, 1.0)               # continues the stream behind the stack's draws
        o.t4o_activate(L, P(Y1), P(Am), P(Fm), a, Y1.size); Fm = Fm.reshape(Y1.shape); cur = Am
    Y2 = np.zeros((N, EB), np.float32); assert o.t4o_linear_fwd(P(np.ascontiguousarray(cur)), P(W2), P(B2), P(Y2), N, EB, EA) == 0
    Pr = np.zeros_like(Y2); o.t4o_softmax(P(Y2), P(Pr), N, EB)
    end = o.t4o_rand_offset()
    arr, bufs = _build(dev, oracle, X, stages, flat, params, ref)
    hd = StackHead()
    d = {"W1": dev.up(W1), "B1": dev.up(B1), "Y1": dev.zeros(Y1.shape), "Fm": dev.zeros(Y1.shape), "Am": dev.zeros(Y1.shape),
         "W2": dev.up(W2), "B2": dev.up(B2), "Y2": dev.zeros(Y2.shape), "P": dev.zeros(Y2.shape)}
    hd.W1, hd.B1, hd.Y1, hd.W2, hd.B2, hd.Y2, hd.P = p(d["W1"]), p(d["B1"]), p(d["Y1"]), p(d["W2"]), p(d["B2"]), p(d["Y2"]), p(d["P"])
    if mid:
        hd.mid_layer, hd.mid_alpha = LAY[mid]; hd.mid_mask, hd.mid_out = p(d["Fm"]), p(d["Am"])
    hd.E1, hd.E0a, hd.E0b = E1, EA, EB
    assert t4k.lib.t4k_conv_stack_head_ok(arr, len(stages), N, ctypes.byref(hd)) == 1
    dX = dev.up(X)
    for rep in range(2):
        t4k.call("t4k_rand_init", seed); t4k.call("t4k_rand_set_offset", off)
        for k_ in ("Y1", "Fm", "Am", "Y2", "P"):
            d[k_].fill_(-7.0)
        t4k.call("t4k_conv_stack_head_fwd", p(dX), None, arr, len(stages), N, ctypes.byref(hd), None)
        assert t4k.lib.t4k_rand_offset() == end
        for si in range(len(stages)):                              # the stack's own tensors, as in the forward test
            for k_ in ("O", "pool_out", "post_out", "copy_out"):
                if k_ in ref[si]:
                    assert rel(dev.down(bufs[si][k_]).reshape(ref[si][k_].shape), ref[si][k_]) < RTOL, "rep %d stage %d %s" % (rep, si, k_)
        assert rel(dev.down(d["Y1"]), Y1) < RTOL, "rep %d linear 1: %.3g" % (rep, rel(dev.down(d["Y1"]), Y1))
        if mid == "dropout":
            assert np.array_equal(dev.down(d["Fm"]), Fm), "rep %d dropout mask" % rep
        elif mid:
            assert np.mean(np.abs(dev.down(d["Fm"]) - Fm) > 1e-3) < 1e-3, "rep %d mask" % rep
        if mid:
            assert rel(dev.down(d["Am"]), Am) < RTOL, "rep %d activation" % rep
        assert rel(dev.down(d["Y2"]), Y2) < RTOL, "rep %d linear 2: %.3g" % (rep, rel(dev.down(d["Y2"]), Y2))
        assert rel(dev.down(d["P"]), Pr) < RTOL, "rep %d softmax: %.3g" % (rep, rel(dev.down(d["P"]), Pr))


def test_lenet_steps_while_another_stream_hogs_the_device_are_deterministic():
    """The one-launch forward (bands meet through epoch-tagged words, the last band polls) and the fused head backward (a counter in front of the
    in-place `out -= target`) hold inter-workgroup waits.  Forty training steps of the LeNet net at batch 128 run twice from the same seed -
    once on an idle device, once while a bandwidth-hogging kernel chain occupies it on another stream (what a concurrent RCCL kernel does):
    no bounded wait may give up (t4k_sync reports that) and every parameter must come out bit-identical."""
    import torch
    from lenet_parity import PARAMS, _get, _setup
    from tensorforth_amd import lib as t4lib
    from tensorforth_amd.vm import VM
    k = t4lib.load()
    res = []
    for hog in (False, True):
        vm = VM(device=0, seed=4242)
        try:
            _setup(vm, 128, 0, 128)
            side = torch.cuda.Stream(); big = torch.zeros(256 << 20 >> 2, device="cuda")
            for _ in range(4):
                if hog:
                    with torch.cuda.stream(side):
                        for _ in range(30):
                            big.mul_(1.0001).add_(1.0)
                vm.eval("net fw bw opt fw bw opt fw bw opt fw bw opt fw bw opt fw bw opt fw bw opt fw bw opt fw bw opt fw bw opt drop\n")
            assert k.lib.t4k_sync(None) == 0, k.lib.t4k_last_error()
            torch.cuda.synchronize()
            res.append({n_: _get(vm, e) for n_, e in PARAMS})
        finally:
            vm.close()
    for n_ in res[0]:
        assert np.isfinite(res[0][n_]).all()
        assert np.array_equal(res[0][n_], res[1][n_]), "%s differs between the idle and the busy device" % n_
