"""Cross-check the parts of the oracle that no reference test pins (conv, pooling, softmax,
batchnorm, optimizers, activations) against torch-CPU, and assert the reference quirks the
oracle must carry (a-11 flipped dX filter, a-14 dropout w/o rescale and SELU
positive branch, a-17 batchnorm eps placement / mean-accumulated dgamma,dbeta, a-19 Adam
without bias correction).  CPU only."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

RTOL = 1e-4     # north_star: fp32 math within 1e-4 relative


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b)) / max(1e-30, np.max(np.abs(b)))


def nhwc(t):    # torch NCHW -> numpy NHWC
    return np.ascontiguousarray(t.permute(0, 2, 3, 1).numpy())


@pytest.mark.parametrize("K,S,P", [(1, 1, 0), (3, 1, 1), (4, 2, 1), (5, 1, 2)])
def test_conv2d_forward_and_backward(oracle, K, S, P):
    o = oracle.lib(); P_ = oracle.P
    rng = np.random.default_rng(K * 10 + S)
    N, H1, C1, C0 = 2, 8, 3, 4
    I = rng.standard_normal((N, H1, H1, C1)).astype(np.float32)
    F = rng.standard_normal((C1, K, K, C0)).astype(np.float32)       # reference layout T4(C1,K,K,C0)
    B = rng.standard_normal(C0).astype(np.float32)
    O = oracle.conv2d_fwd(I, F, B, K, S, P)
    H0 = O.shape[1]
    ti = torch.tensor(I).permute(0, 3, 1, 2).requires_grad_(True)
    tw = torch.tensor(F).permute(3, 0, 1, 2).contiguous().requires_grad_(True)   # [C0,C1,K,K]
    tb = torch.tensor(B).requires_grad_(True)
    to = Fn.conv2d(ti, tw, tb, stride=S, padding=P)
    assert rel(O, nhwc(to.detach())) < RTOL
    dO = rng.standard_normal(O.shape).astype(np.float32)
    to.backward(torch.tensor(dO).permute(0, 3, 1, 2))
    DX = np.zeros_like(I); DF = np.zeros_like(F); DB = np.zeros_like(B)
    rc = o.t4o_conv2d_bwd(P_(I), P_(dO), P_(DX), P_(F), P_(DF), P_(DB), N, H1, H1, C1, H0, H0, C0, K, S, P, 1)
    assert rc == 0
    assert rel(DF, tw.grad.permute(1, 2, 3, 0).numpy()) < RTOL        # dF textbook
    assert rel(DB, tb.grad.numpy()) < RTOL
    # quirk a-11: dX is the textbook gradient of a conv whose filter is rotated by 180 degrees
    tw2 = torch.flip(tw.detach(), dims=(2, 3)).requires_grad_(False)
    ti2 = torch.tensor(I).permute(0, 3, 1, 2).requires_grad_(True)
    Fn.conv2d(ti2, tw2, None, stride=S, padding=P).backward(torch.tensor(dO).permute(0, 3, 1, 2))
    assert rel(DX, nhwc(ti2.grad)) < RTOL
    if K > 1:
        assert rel(DX, nhwc(ti.grad)) > 1e-2                           # ... and NOT the textbook one
    # accumulation semantics: a second call adds to DF/DB, overwrites DX
    o.t4o_conv2d_bwd(P_(I), P_(dO), P_(DX), P_(F), P_(DF), P_(DB), N, H1, H1, C1, H0, H0, C0, K, S, P, 1)
    assert rel(DF, 2 * tw.grad.permute(1, 2, 3, 0).numpy()) < RTOL
    assert rel(DX, nhwc(ti2.grad)) < RTOL
    assert o.t4o_conv2d_fwd(P_(I), P_(O), P_(F), P_(B), N, H1, H1, C1, H0, H0, C0, 7, 1, 3) == -4   # unsupported


@pytest.mark.parametrize("KS", [2, 3])
def test_pooling(oracle, KS):
    o = oracle.lib(); P_ = oracle.P
    rng = np.random.default_rng(KS)
    N, H1, C = 2, 6 * KS, 3
    I = rng.standard_normal((N, H1, H1, C)).astype(np.float32)
    H0 = H1 // KS
    ti = torch.tensor(I).permute(0, 3, 1, 2).requires_grad_(True)
    for layer, tf in ((oracle.L_MAXPOOL, Fn.max_pool2d), (oracle.L_AVGPOOL, Fn.avg_pool2d)):
        O = np.zeros((N, H0, H0, C), np.float32)
        assert o.t4o_pool(layer, P_(I), P_(O), N, H1, H1, H0, H0, C, KS) == 0
        to = tf(ti, KS)
        assert np.array_equal(O, nhwc(to.detach())) or rel(O, nhwc(to.detach())) < 1e-6
        dY = rng.standard_normal(O.shape).astype(np.float32)
        ti.grad = None
        to.backward(torch.tensor(dY).permute(0, 3, 1, 2))
        X = I.copy()
        assert o.t4o_dpool(layer, P_(X), P_(dY), N, H1, H1, H0, H0, C, KS) == 0     # in place on the input
        assert rel(X, nhwc(ti.grad)) < 1e-6
    O = np.zeros((N, H0, H0, C), np.float32)
    o.t4o_pool(oracle.L_MINPOOL, P_(I), P_(O), N, H1, H1, H0, H0, C, KS)
    assert np.array_equal(O, -nhwc(Fn.max_pool2d(-ti.detach(), KS)))


def test_softmax_and_ce_loss(oracle):
    o = oracle.lib(); P_ = oracle.P
    rng = np.random.default_rng(3)
    for C in (10, 300):
        x = (rng.standard_normal((7, C)) * 3).astype(np.float32); y = np.zeros_like(x)
        o.t4o_softmax(P_(x), P_(y), 7, C)
        assert rel(y, torch.softmax(torch.tensor(x), 1).numpy()) < RTOL
        assert np.allclose(y.sum(1), 1, atol=1e-5)


def test_batchnorm_quirks(oracle):
    o = oracle.lib(); P_ = oracle.P
    rng = np.random.default_rng(4)
    N, H, W, C = 4, 5, 5, 3; HW = H * W; NHW = N * HW
    x = (rng.standard_normal((N, H, W, C)) * 2 + 1).astype(np.float32)
    g = rng.standard_normal(C).astype(np.float32); b = rng.standard_normal(C).astype(np.float32)
    y = np.zeros_like(x); xh = np.zeros_like(x); stat = np.zeros(3 * C, np.float32)
    o.t4o_batchnorm_fwd(P_(x), P_(y), P_(xh), P_(g), P_(b), P_(stat), N, HW, C)
    tx = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_(True)
    tg = torch.tensor(g).requires_grad_(True); tb = torch.tensor(b).requires_grad_(True)
    ty = Fn.batch_norm(tx, None, None, tg, tb, training=True, eps=1e-12)
    assert rel(y, nhwc(ty.detach())) < RTOL                         # eps=1e-6 outside sqrt ~ no eps at var~4
    var = x.reshape(-1, C).var(0)
    assert np.allclose(stat[:C], 1.0 / (np.sqrt(var) + 1e-6), rtol=1e-4)     # a-17: eps outside sqrt
    dy = rng.standard_normal(x.shape).astype(np.float32)
    ty.backward(torch.tensor(dy).permute(0, 3, 1, 2))
    dx = np.zeros_like(x); dw = np.zeros(C, np.float32); db = np.zeros(C, np.float32)
    o.t4o_batchnorm_bwd(P_(g), P_(dy), P_(xh), P_(dx), P_(dw), P_(db), P_(stat), N, HW, C, 1)
    assert rel(dx, nhwc(tx.grad)) < 5e-4
    assert rel(dw, tg.grad.numpy() / NHW) < RTOL                    # a-17: the MEANS are accumulated
    assert rel(db, tb.grad.numpy() / NHW) < RTOL


def test_optimizers(oracle):
    o = oracle.lib(); P_ = oracle.P
    rng = np.random.default_rng(5)
    n = 1000
    w = rng.standard_normal(n).astype(np.float32); g = rng.standard_normal(n).astype(np.float32)
    m = rng.standard_normal(n).astype(np.float32) * 0.1; v = np.abs(rng.standard_normal(n)).astype(np.float32) * 0.1
    # Adam, a-19: no bias correction, eps after sqrt
    w1, g1, m1, v1 = w.copy(), g.copy(), m.copy(), v.copy()
    o.t4o_adam(P_(w1), P_(g1), P_(m1), P_(v1), 1e-3, 0.9, 0.999, n)
    em = 0.9 * m + 0.1 * g; ev = 0.999 * v + 0.001 * g * g
    assert rel(m1, em) < 1e-6 and rel(v1, ev) < 1e-5
    assert rel(w1, w - 1e-3 * em / (np.sqrt(ev) + 1e-6)) < 1e-6
    assert not g1.any()
    # SGD plain (beta ~ 0) divides by Nw (parameter tensor's N), not by batch
    w2, g2 = w.copy(), g.copy()
    o.t4o_sgd(P_(w2), P_(g2), P_(w2), 3, 0.5, 0.0, n)
    assert rel(w2, w - 0.5 * g / 3) < 1e-6 and not g2.any()
    # SGD momentum
    w3, g3, m3 = w.copy(), g.copy(), m.copy()
    o.t4o_sgd(P_(w3), P_(g3), P_(m3), 1, 0.1, 0.9, n)
    assert rel(m3, 0.9 * m + 0.1 * g) < 1e-6 and rel(w3, w - 0.1 * (0.9 * m + 0.1 * g)) < 1e-6
    # AdamW
    w4, g4, m4, v4 = w.copy(), g.copy(), m.copy(), v.copy()
    o.t4o_adamw(P_(w4), P_(g4), P_(m4), P_(v4), 1e-3, 0.9, 0.999, 0.01, n)
    assert rel(w4, w - 1e-3 * (em / (np.sqrt(ev) + 1e-6) - 0.01 * g)) < 1e-6


def test_activations(oracle):
    o = oracle.lib(); P_ = oracle.P
    x = np.linspace(-3, 3, 61).astype(np.float32); tx = torch.tensor(x)
    def act(layer, alpha=0.0, f_in=None):
        y = np.zeros_like(x); f = np.zeros_like(x) if f_in is None else f_in.copy()
        assert o.t4o_activate(layer, P_(x), P_(y), P_(f), alpha, x.size) == 0
        return y, f
    y, f = act(oracle.L_RELU); assert np.array_equal(y, np.maximum(x, 0)) and np.array_equal(f, (x > 0).astype(np.float32))
    y, f = act(oracle.L_TANH); assert rel(y, np.tanh(x)) < 1e-6 and rel(f, 1 - np.tanh(x) ** 2) < 1e-5
    y, f = act(oracle.L_SIGMOID); s = torch.sigmoid(tx).numpy(); assert rel(y, s) < 1e-6 and rel(f, s * (1 - s)) < 1e-5
    y, f = act(oracle.L_LEAKYRL, 0.01); assert rel(y, Fn.leaky_relu(tx, 0.01).numpy()) < 1e-6
    y, f = act(oracle.L_ELU, 1.0); assert rel(y, Fn.elu(tx, 1.0).numpy()) < 1e-6
    # SELU quirk: negative branch matches torch's selu, positive branch returns x (not 1.0507*x)
    y, f = act(oracle.L_SELU); ts = Fn.selu(tx).numpy()
    assert rel(y[x <= 0], ts[x <= 0]) < 2e-4 and np.array_equal(y[x > 0], x[x > 0])
    assert np.allclose(f[x > 0], 1.0507)
    # dropout: mask = rand > p, survivors are NOT rescaled by 1/(1-p)
    r = np.random.default_rng(6).random(x.size).astype(np.float32)
    y, f = act(oracle.L_DROPOUT, 0.5, r)
    assert np.array_equal(f, (r > 0.5).astype(np.float32)) and np.array_equal(y, x * f)


def test_gemm_variants_and_misc(oracle):
    o = oracle.lib(); P_ = oracle.P
    rng = np.random.default_rng(7)
    M, N, K = 5, 7, 9
    A = rng.standard_normal((M, K)).astype(np.float32); B = rng.standard_normal((K, N)).astype(np.float32)
    O0 = rng.standard_normal((M, N)).astype(np.float32)
    ref = 0.5 * A.astype(np.float64) @ B + 2.0 * O0
    for tA in (0, 1):
        for tB in (0, 1):
            As = np.ascontiguousarray(A.T) if tA else A; Bs = np.ascontiguousarray(B.T) if tB else B
            assert rel(oracle.gemm(As, Bs, O0.copy(), 0.5, 2.0, tA, tB), ref) < 1e-6
    # channel-interleaved (C=2): two independent GEMMs
    A2 = rng.standard_normal((M, K, 2)).astype(np.float32); B2 = rng.standard_normal((K, N, 2)).astype(np.float32)
    O2 = oracle.gemm(A2, B2, C=2)
    for c in range(2):
        assert rel(O2[..., c], A2[..., c].astype(np.float64) @ B2[..., c]) < 1e-6
    # the reference's host GEMM (word `gemm`) agrees
    O3 = O0.copy(); o.t4o_gemm_host_blocked(P_(A), P_(B), P_(O3), 0.5, 2.0, M, N, K)
    assert rel(O3, ref) < 1e-6
    # transpose / identity are exact
    T = np.zeros((K, M), np.float32); o.t4o_transpose(P_(A), P_(T), M, K, 1); assert np.array_equal(T, A.T)
    E = np.ones((4, 4), np.float32); o.t4o_identity(P_(E), 4, 4, 1); assert np.array_equal(E, np.eye(4, dtype=np.float32))
    # std() = sqrt(sum (x-avg)^2)/numel (tensor.cu:242-250 quirk): n*var reduction
    x = rng.standard_normal(5000).astype(np.float32)
    assert abs(oracle.reduce(oracle.RED_NVAR, x, float(x.mean())) - ((x - x.mean()) ** 2).sum()) < 0.05
    assert abs(oracle.reduce(oracle.RED_SUM, x) - x.astype(np.float64).sum()) < 1e-2
    assert oracle.reduce(oracle.RED_MAX, x) == x.max() and oracle.reduce(oracle.RED_MIN, x) == x.min()
    # hit / onehot: label >= classes maps to class 0 (loss.cpp:66); first max wins
    lab = np.array([1, 12, 0], np.uint32); hot = np.zeros((3, 3), np.float32); o.t4o_onehot(P_(lab), P_(hot), 3, 3)
    assert np.array_equal(hot, np.array([[0, 1, 0], [1, 0, 0], [1, 0, 0]], np.float32))
    out = np.array([[0, 5, 5], [9, 1, 1], [2, 2, 1]], np.float32); cnt = ctypes.c_int(0)
    o.t4o_hit(P_(out), P_(hot), 3, 3, ctypes.byref(cnt)); assert cnt.value == 3


def test_philox_stream_properties(oracle):
    o = oracle.lib(); P_ = oracle.P
    o.t4o_rand_init(42)
    a = np.zeros(10001, np.float32); o.t4o_rand(P_(a), a.size, 0, 0.0, 1.0)
    assert a.min() > 0.0 and a.max() <= 1.0 and abs(a.mean() - 0.5) < 0.02      # uniform (0,1]
    assert o.t4o_rand_offset() == 10004
    b = np.zeros(10001, np.float32); o.t4o_rand(P_(b), b.size, 1, 0.0, 1.0)
    assert abs(b.mean()) < 0.05 and abs(b.std() - 1.0) < 0.05                    # N(0,1)
    o.t4o_rand_init(42); c = np.zeros(10001, np.float32); o.t4o_rand(P_(c), c.size, 0, 0.0, 1.0)
    assert np.array_equal(a, c)                                                   # reproducible
    # Philox4x32-10 known-answer (Random123 kat_vectors: ctr=0,key=0)
    o.t4o_rand_init(0); d = np.zeros(4, np.float32); o.t4o_rand(P_(d), 4, 0, 0.0, 1.0)
    kat = np.array([0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8], np.uint64)
    exp = (kat.astype(np.float32) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33)).astype(np.float32)
    assert np.allclose(d, exp, rtol=0, atol=1e-7)


@pytest.mark.parametrize("H1", [4, 7])                  # even: H0 = 2*H1; odd: the reference's output padding P0 = 1 gives H0 = 2*H1 + 1
def test_transposed_conv_layer_is_torch_conv_transpose2d(oracle, H1):
    """Word `dconv2d` (L_DCONV, K=4 S=2 P=1; allocation Model::_iconv txn, model.cpp:121-180).  The reference never finished the layer
    (forward.cu:110 / backprop.cu:137 call the conv routines with the operands unswapped); the oracle states the finished one:
    torch ConvTranspose2d with weight[ci][co][ky][kx] = F[ci,ky,kx,co] (+ output padding P0), gradients textbook, DF/DB accumulating."""
    o = oracle.lib(); P_ = oracle.P
    K, S, P = 4, 2, 1
    rng = np.random.default_rng(H1)
    N, C1, C0 = 2, 5, 3
    P0 = (H1 + 2 * P - K) % S
    H0 = (H1 - 1) * S - 2 * P + K + P0
    I = rng.standard_normal((N, H1, H1, C1)).astype(np.float32)
    F = rng.standard_normal((C1, K, K, C0)).astype(np.float32)
    B = rng.standard_normal(C0).astype(np.float32)
    O = np.zeros((N, H0, H0, C0), np.float32)
    assert o.t4o_dconv2d_fwd(P_(I), P_(O), P_(F), P_(B), N, H1, H1, C1, H0, H0, C0, K, S, P) == 0
    ti = torch.tensor(I).permute(0, 3, 1, 2).requires_grad_(True)
    tw = torch.tensor(F).permute(0, 3, 1, 2).contiguous().requires_grad_(True)     # [C1, C0, K, K]
    tb = torch.tensor(B).requires_grad_(True)
    to = Fn.conv_transpose2d(ti, tw, tb, stride=S, padding=P, output_padding=P0)
    assert tuple(to.shape[2:]) == (H0, H0)
    assert rel(O, nhwc(to.detach())) < RTOL
    dO = rng.standard_normal(O.shape).astype(np.float32)
    to.backward(torch.tensor(dO).permute(0, 3, 1, 2))
    DX = np.full_like(I, 7.0); DF = np.zeros_like(F); DB = np.zeros_like(B)          # DX is overwritten, DF / DB accumulate
    for rep in (1, 2):
        assert o.t4o_dconv2d_bwd(P_(I), P_(dO), P_(DX), P_(F), P_(DF), P_(DB), N, H1, H1, C1, H0, H0, C0, K, S, P, 1) == 0
        assert rel(DX, nhwc(ti.grad)) < RTOL
        assert rel(DF, rep * tw.grad.permute(0, 2, 3, 1).numpy()) < RTOL
        assert rel(DB, rep * tb.grad.numpy()) < RTOL
