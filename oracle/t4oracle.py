"""Python face of the CPU oracle (TEST INFRASTRUCTURE ONLY - see t4_oracle.h).

* ``lib()``            - ctypes handle on oracle/libt4oracle.so (built on demand with make)
* thin numpy wrappers  - ``gemm``, ``conv2d_fwd`` ... operating on float32 ndarrays
* ``OracleModel``      - restatement of the reference's host orchestration
  (Model::add / forward / backprop / sgd / adam / loss: src/nn/model.cpp:82-310,
  src/nn/forward.cu:28-113, src/nn/backprop.cu:39-140, src/nn/gradient.cu:63-169,
  src/nn/loss.cpp:119-136, src/mu/tensor.cu:288-325) on top of the oracle kernels, used to
  pin the oracle against the reference's known-answer scripts and as the CPU baseline.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes
import math
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from tensorforth_amd import _cabi  # noqa: E402  (header parser only; no product code path)

_lib = None

# enums (include/t4k.h == reference src/t4math.h:25-56, src/nn/ntypes.h:16-36)
ABS, NEG, EXP, LN, LOG, TANH, RELU, SIGM, SQRT, RCP, SAT, IDEN, FILL, GFILL, SCALE, POW, ADD, SUB, MUL, DIV = range(20)
(L_NONE, L_CONV, L_LINEAR, L_FLATTEN, L_RELU, L_TANH, L_SIGMOID, L_SELU, L_LEAKYRL, L_ELU, L_DROPOUT,
 L_SOFTMAX, L_LOGSMAX, L_AVGPOOL, L_MAXPOOL, L_MINPOOL, L_BATCHNM, L_USAMPLE, L_DCONV) = range(19)
RED_SUM, RED_NVAR, RED_MAX, RED_MIN = range(4)
LOSS_MSE, LOSS_BCE, LOSS_CE, LOSS_NLL = range(4)
LAYER_NAMES = ["output ", "conv2d ", "linear ", "flatten", "relu   ", "tanh   ", "sigmoid", "selu   ", "leakyrl",
               "elu    ", "dropout", "softmax", "logsmax", "avgpool", "maxpool", "minpool", "batchnm", "upsampl", "dconv2d"]


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "libt4oracle.so")
        src = os.path.join(_HERE, "t4_oracle.cpp")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib = ctypes.CDLL(so)
        decls = _cabi.parse_header(os.path.join(_HERE, "t4_oracle.h"), "t4o_")
        missing = _cabi.bind(_lib, decls)
        assert not missing, missing
    return _lib


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def P(a):
    return a.ctypes.data if a is not None else None


def scalar_lsb(v):
    """SCALAR(): clear the mantissa LSB of an fp32 (reference src/t4base.h:27,30)."""
    u = np.float32(v).view(np.uint32) & np.uint32(0xFFFFFFFE)
    return float(u.view(np.float32))


# ----------------------------------------------------------------------------- wrappers
def reduce(op, x, avg=0.0):
    x = f32(x).ravel(); out = np.zeros(1, np.float32)
    assert lib().t4o_reduce(op, P(x), x.size, avg, P(out)) == 0
    return float(out[0])


def gemm(A, B, O=None, alpha=1.0, beta=0.0, tA=0, tB=0, C=1):
    """A, B as stored (row-major [rows, cols(, C)]); returns O[M,N(,C)]."""
    A = f32(A); B = f32(B)
    a_r, a_c = A.shape[0], A.shape[1]; b_r, b_c = B.shape[0], B.shape[1]
    M, K = (a_c, a_r) if tA else (a_r, a_c)
    N = b_r if tB else b_c
    shape = (M, N) if C == 1 else (M, N, C)
    O = np.zeros(shape, np.float32) if O is None else f32(O)
    assert lib().t4o_gemm(P(A), P(B), P(O), alpha, beta, tA, tB, M, N, K, C) == 0
    return O


def conv2d_fwd(I, F, B, K, S, P_):
    N, H1, W1, C1 = I.shape; C0 = F.shape[-1]
    H0 = (H1 - K + 2 * P_) // S + 1; W0 = (W1 - K + 2 * P_) // S + 1
    O = np.zeros((N, H0, W0, C0), np.float32)
    rc = lib().t4o_conv2d_fwd(P(f32(I)), P(O), P(f32(F)), P(f32(B)), N, H1, W1, C1, H0, W0, C0, K, S, P_)
    assert rc == 0, rc
    return O


# ------------------------------------------------------------------------------ model
class Layer:
    def __init__(self, fn, x):
        self.fn = fn          # t4_layer of the op applied to x
        self.x = x            # layer input tensor (NHWC ndarray); becomes dX after backprop
        self.w = self.b = self.dw = self.db = self.aux = None
        self.m = [None, None, None, None]   # mtum[0..3]: m_w, m_b, v_w, v_b
        self.stat = None      # batchnorm mtum[4]
        self.stride = 1; self.pad = 0; self.xparm = 0.0; self.k = 0


class OracleModel:
    """Restates nn::Model on numpy arrays + oracle kernels.  Layer i consumes tensors[i] and
    produces tensors[i+1]; gradients overwrite the forward activations in place
    (src/nn/backprop.cu:111-140)."""

    def __init__(self, n, h, w, c, seed=1234):
        self.t = [np.zeros((n, h, w, c), np.float32)]
        self.layers = []
        self.train = True
        self.iter = 0
        self.epoch = 0
        self.hot = None
        self.hit = 0
        lib().t4o_rand_init(seed)

    # -- random init: Model::RAND src/nn/model.cpp:73-78 -> 2k*(u-0.5)
    def _rand(self, shape, scale):
        a = np.zeros(shape, np.float32)
        lib().t4o_rand(P(a), a.size, 0, -0.5, float(np.float32(scale * 2.0)))
        return a

    def _push(self, fn, out_shape):
        L = Layer(fn, None)
        self.layers.append(L)
        self.t.append(np.zeros(out_shape, np.float32))
        return L

    def conv2d(self, c0, bias=0.5, k=3, s=1, p=None):           # _iconv src/nn/model.cpp:121-180
        n, h1, w1, c1 = self.t[-1].shape
        if p is None:
            p = (k - 1) // 2
        h0 = (h1 - k + 2 * p) // s + 1
        w0 = (h1 - k + 2 * p) // s + 1                           # W0 from H1: reference quirk :137
        L = self._push(L_CONV, (n, h0, w0, c0))
        L.k, L.stride, L.pad, L.xparm = k, s, p, bias
        kk = math.sqrt(6.0 / (k * k * c1))
        L.w = self._rand((c1, k, k, c0), kk); L.b = self._rand((c0,), bias)
        L.dw = np.zeros_like(L.w); L.db = np.zeros_like(L.b)
        L.aux = np.zeros((n, h1, w1, c1), np.float32)            # dx
        return self

    def linear(self, e0, bias=1.0):                              # _ilinear src/nn/model.cpp:182-226
        n = self.t[-1].shape[0]; e1 = int(np.prod(self.t[-1].shape[1:]))
        L = self._push(L_LINEAR, (n, 1, e0, 1))
        L.xparm = bias
        kk = math.sqrt(1.0 / (e0 + e1))
        L.w = self._rand((e0, e1), kk); L.b = self._rand((e0,), bias)
        L.dw = np.zeros_like(L.w); L.db = np.zeros_like(L.b)
        return self

    def flatten(self):
        n = self.t[-1].shape[0]; e = int(np.prod(self.t[-1].shape[1:]))
        self._push(L_FLATTEN, (n, 1, e, 1)); return self

    def activate(self, fn, alpha=0.0):                           # _iactivate :247-256
        L = self._push(fn, self.t[-1].shape)
        L.aux = np.zeros(self.t[-2].shape, np.float32); L.xparm = alpha
        return self

    def relu(self): return self.activate(L_RELU)
    def tanh(self): return self.activate(L_TANH)
    def sigmoid(self): return self.activate(L_SIGMOID)
    def selu(self): return self.activate(L_SELU)
    def leakyrelu(self, a=0.01): return self.activate(L_LEAKYRL, a)
    def elu(self, a=1.0): return self.activate(L_ELU, a)
    def dropout(self, p): return self.activate(L_DROPOUT, p)

    def softmax(self):
        self._push(L_SOFTMAX, self.t[-1].shape); return self

    def pool(self, fn, k):                                       # _ipool :260-274 (ceil dims)
        n, h, w, c = self.t[-1].shape
        L = self._push(fn, (n, (h + k - 1) // k, (w + k - 1) // k, c)); L.stride = k
        return self

    def maxpool(self, k): return self.pool(L_MAXPOOL, k)
    def avgpool(self, k): return self.pool(L_AVGPOOL, k)
    def minpool(self, k): return self.pool(L_MINPOOL, k)

    def batchnorm(self, m=0.1):                                  # _ibatchnorm :276-292
        c = self.t[-1].shape[3]
        L = self._push(L_BATCHNM, self.t[-1].shape)
        L.w = np.ones(c, np.float32); L.b = np.zeros(c, np.float32)
        L.dw = np.zeros(c, np.float32); L.db = np.zeros(c, np.float32)
        L.aux = np.zeros(self.t[-2].shape, np.float32); L.stat = np.zeros(3 * c, np.float32); L.xparm = m
        return self

    # -- forward: Model::forward / _fstep src/nn/forward.cu:28-113
    def forward(self, x):
        o = lib()
        x = f32(x)
        assert x.size == self.t[0].size
        self.t[0][...] = x.reshape(self.t[0].shape)              # n0 = input (copy)
        for i, L in enumerate(self.layers):
            a, y = self.t[i], self.t[i + 1]
            fn = L.fn
            if fn == L_CONV:
                n, h1, w1, c1 = a.shape; _, h0, w0, c0 = y.shape
                rc = o.t4o_conv2d_fwd(P(a), P(y), P(L.w), P(L.b), n, h1, w1, c1, h0, w0, c0, L.k, L.stride, L.pad)
                assert rc == 0
            elif fn == L_LINEAR:
                n = a.shape[0]; e1 = a.size // n; e0 = y.size // n
                o.t4o_linear_fwd(P(a), P(L.w), P(L.b), P(y), n, e0, e1)
            elif fn == L_FLATTEN:
                y.ravel()[:] = a.ravel()
            elif fn in (L_RELU, L_TANH, L_SIGMOID, L_SELU, L_LEAKYRL, L_ELU, L_DROPOUT):
                if fn == L_DROPOUT:
                    o.t4o_dropout_mask(P(L.aux), L.aux.size)       # keyed by sample when a data-parallel shard is set
                o.t4o_activate(fn, P(a), P(y), P(L.aux), L.xparm, a.size)
            elif fn == L_SOFTMAX:
                n = a.shape[0]; o.t4o_softmax(P(a), P(y), n, a.size // n)
            elif fn in (L_AVGPOOL, L_MAXPOOL, L_MINPOOL):
                n, h1, w1, c = a.shape; _, h0, w0, _ = y.shape
                o.t4o_pool(fn, P(a), P(y), n, h1, w1, h0, w0, c, L.stride)
            elif fn == L_BATCHNM:
                n, h, w, c = a.shape
                o.t4o_batchnorm_fwd(P(a), P(y), P(L.aux), P(L.w), P(L.b), P(L.stat), n, h * w, c)
            else:
                raise NotImplementedError(fn)
        return self.t[-1]

    def onehot_labels(self, labels):
        out = self.t[-1]; n = out.shape[0]; e = out.size // n
        lab = np.ascontiguousarray(labels, dtype=np.uint32)
        self.hot = np.zeros((n, 1, e, 1), np.float32)
        lib().t4o_onehot(P(lab), P(self.hot), n, e)
        cnt = ctypes.c_int(0)
        lib().t4o_hit(P(self.t[-1]), P(self.hot), n, e, ctypes.byref(cnt))
        self.hit = cnt.value
        return self.hot

    # -- loss: Model::loss src/nn/loss.cpp:119-136 + Tensor::loss src/mu/tensor.cu:288-325
    def loss(self, op, tgt):
        o = lib()
        out = self.t[-1].copy().ravel(); t = f32(tgt).ravel()
        assert out.size == t.size
        n = self.t[-1].shape[0]
        if op == LOSS_MSE:
            o.t4o_tt_op(SUB, P(out), P(t), P(out), out.size)
            o.t4o_tt_op(MUL, P(out), P(out), P(out), out.size)
            z = self._sum(out)
        elif op == LOSS_BCE:
            r = np.zeros(1, np.float32); o.t4o_bce(P(t), P(out), out.size, P(r)); z = -float(r[0])
        else:
            if op == LOSS_CE:
                o.t4o_math(LN, P(out), 0.0, out.size)
            o.t4o_tt_op(MUL, P(out), P(t), P(out), out.size)
            z = -self._sum(out)
        z = np.float32(z) / np.float32(n)
        return scalar_lsb(z)

    @staticmethod
    def _sum(a):                                                 # Tensor::sum tensor.cu:224-236
        if a.size < 16:
            v = np.float32(0)
            for x in a.ravel():
                v = np.float32(v + x)
            return scalar_lsb(v)
        return scalar_lsb(reduce(RED_SUM, a))

    # -- backprop: src/nn/backprop.cu:39-140
    def backprop(self, tgt=None):
        o = lib()
        tgt = self.hot if tgt is None else f32(tgt)
        out = self.t[-1]
        assert out.size == tgt.size
        last_fn = self.layers[-1].fn
        if last_fn in (L_LINEAR, L_SIGMOID, L_SOFTMAX, L_LOGSMAX):     # _bprep :97-103
            o.t4o_tt_op(SUB, P(out), P(f32(tgt)), P(out), out.size)
        else:
            out.ravel()[:] = f32(tgt).ravel()
        nl = len(self.layers)
        for j, i in enumerate(range(nl - 1, -1, -1)):
            L = self.layers[i]; a, y = self.t[i], self.t[i + 1]
            fn = L.fn
            if fn == L_CONV:
                n, h1, w1, c1 = a.shape; _, h0, w0, c0 = y.shape
                rc = o.t4o_conv2d_bwd(P(a), P(y), P(L.aux), P(L.w), P(L.dw), P(L.db),
                                      n, h1, w1, c1, h0, w0, c0, L.k, L.stride, L.pad, int(self.train))
                assert rc == 0
                a[...] = L.aux                                   # in = dx
            elif fn == L_LINEAR:
                if j == 0:
                    a.ravel()[:] = y.ravel()                     # last layer: in = out (:119-121)
                else:
                    n = a.shape[0]; e1 = a.size // n; e0 = y.size // n
                    o.t4o_linear_bwd(P(a), P(L.w), P(y), P(a), P(L.dw), P(L.db), n, e0, e1, int(self.train))
            elif fn in (L_FLATTEN, L_SIGMOID, L_SOFTMAX, L_LOGSMAX):
                a.ravel()[:] = y.ravel()                         # pass-through (:122,129-131)
            elif fn in (L_RELU, L_TANH, L_SELU, L_LEAKYRL, L_ELU, L_DROPOUT):
                o.t4o_tt_op(MUL, P(y), P(L.aux), P(a), a.size)   # in = out * mask
            elif fn in (L_AVGPOOL, L_MAXPOOL, L_MINPOOL):
                n, h1, w1, c = a.shape; _, h0, w0, _ = y.shape
                o.t4o_dpool(fn, P(a), P(y), n, h1, w1, h0, w0, c, L.stride)
            elif fn == L_BATCHNM:
                n, h, w, c = a.shape
                o.t4o_batchnorm_bwd(P(L.w), P(y), P(L.aux), P(a), P(L.dw), P(L.db), P(L.stat), n, h * w, c, int(self.train))
            else:
                raise NotImplementedError(fn)
        return self

    # -- optimizers: src/nn/gradient.cu:63-169
    def _params(self):
        for L in self.layers:
            if L.w is not None and L.dw is not None:
                nw = L.w.shape[0] if L.fn == L_CONV else 1       # g.N(): C1 for conv filters T4(C1,K,K,C0)
                yield L, 0, L.w, L.dw, nw
                yield L, 1, L.b, L.db, 1

    def sgd(self, lr, beta=0.9):
        o = lib()
        b = beta if self.iter else 0.0                           # `_iter ? b : 0` :139
        first = (self.iter == 0 and self.epoch == 0); self.iter += 1
        if not self.train:
            return self
        for L, k, g, dg, nw in self._params():
            if abs(beta) >= 1e-6 and L.m[k] is None and first:
                L.m[k] = np.zeros_like(g)
            m = L.m[k] if L.m[k] is not None else g
            o.t4o_sgd(P(g), P(dg), P(m), nw, lr, b, g.size)
        return self

    def adam(self, lr, b1=0.9, b2=0.999):
        o = lib()
        self.iter += 1
        if not self.train:
            return self
        for L, k, g, dg, nw in self._params():
            if L.m[k] is None:
                L.m[k] = np.zeros_like(g); L.m[k + 2] = np.zeros_like(g)
            o.t4o_adam(P(g), P(dg), P(L.m[k]), P(L.m[k + 2]), lr, b1, b2, g.size)
        return self

    def nparams(self):
        return sum(g.size for _, _, g, _, _ in self._params())
