"""Python mirror of the reference's nn::Model orchestration over the C-ABI (libt4hip.so).

Same layer words / argument meaning as the Forth surface (src/vm/netvm.cpp:292-485) and the
same forward / backprop / optimizer sequencing as src/nn/forward.cu:28-113,
src/nn/backprop.cu:39-140, src/nn/gradient.cu:63-169 - but every tensor lives in HBM and
every op is a t4k_* launch on one stream; the host synchronises only when it reads a value
(loss, hit).  Used by the parity tests and bench.py; the C++ VM (`ten4`) carries the same
logic natively.  torch is used for device memory only.
"""
import ctypes
import math
import struct

import numpy as np

from .lib import load

(L_NONE, L_CONV, L_LINEAR, L_FLATTEN, L_RELU, L_TANH, L_SIGMOID, L_SELU, L_LEAKYRL, L_ELU, L_DROPOUT,
 L_SOFTMAX, L_LOGSMAX, L_AVGPOOL, L_MAXPOOL, L_MINPOOL, L_BATCHNM, L_USAMPLE, L_DCONV) = range(19)
ADD, SUB, MUL, DIV = 16, 17, 18, 19
LN = 3
RED_SUM = 0
LOSS_MSE, LOSS_BCE, LOSS_CE, LOSS_NLL = range(4)


def _p(t):
    return t.data_ptr() if t is not None else None


class _Layer:
    def __init__(self, fn):
        self.fn = fn
        self.w = self.b = self.dw = self.db = self.aux = self.stat = None
        self.m = [None, None, None, None]
        self.stride = 1; self.pad = 0; self.xparm = 0.0; self.k = 0


class Model:
    def __init__(self, n, h, w, c, seed=1234, device="cuda:0", stream=None):
        import torch
        self.torch = torch
        self.dev = torch.device(device)
        self.k = load()
        self.k.init(self.dev.index or 0)
        self.s = stream                      # None -> library default stream
        if stream is None:                   # share torch's (null) stream: torch allocs/copies and
            self.k.call("t4k_set_default_stream", None)   # RCCL collectives order with our launches
        self.t = [self._zeros((n, h, w, c))]
        self.layers = []
        self.train = True
        self.iter = 0
        self.epoch = 0
        self.hot = None
        self._hit = self.torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._scalar = self._zeros(1)
        self._graphs = {}
        self.k.call("t4k_rand_init", seed)

    # ---- memory helpers
    def _zeros(self, shape):
        return self.torch.zeros(shape, dtype=self.torch.float32, device=self.dev)

    def _rand(self, shape, scale):           # Model::RAND src/nn/model.cpp:73-78
        a = self._zeros(shape)
        self.k.call("t4k_rand", _p(a), a.numel(), 0, -0.5, float(np.float32(scale * 2.0)), self.s)
        return a

    def _push(self, fn, out_shape):
        L = _Layer(fn); self.layers.append(L); self.t.append(self._zeros(out_shape)); return L

    # ---- layer words
    def conv2d(self, c0, bias=0.5, k=3, s=1, p=None):
        n, h1, w1, c1 = self.t[-1].shape
        if p is None:
            p = (k - 1) // 2
        h0 = (h1 - k + 2 * p) // s + 1
        L = self._push(L_CONV, (n, h0, h0, c0))
        L.k, L.stride, L.pad, L.xparm = k, s, p, bias
        L.w = self._rand((c1, k, k, c0), math.sqrt(6.0 / (k * k * c1))); L.b = self._rand((c0,), bias)
        L.dw = self._zeros(L.w.shape); L.db = self._zeros(L.b.shape); L.aux = self._zeros((n, h1, w1, c1))
        return self

    def linear(self, e0, bias=1.0):
        n = self.t[-1].shape[0]; e1 = self.t[-1].numel() // n
        L = self._push(L_LINEAR, (n, 1, e0, 1)); L.xparm = bias
        L.w = self._rand((e0, e1), math.sqrt(1.0 / (e0 + e1))); L.b = self._rand((e0,), bias)
        L.dw = self._zeros(L.w.shape); L.db = self._zeros(L.b.shape)
        return self

    def flatten(self):
        n = self.t[-1].shape[0]; self._push(L_FLATTEN, (n, 1, self.t[-1].numel() // n, 1)); return self

    def _act(self, fn, alpha=0.0):
        L = self._push(fn, tuple(self.t[-1].shape)); L.aux = self._zeros(tuple(self.t[-2].shape)); L.xparm = alpha
        return self

    def relu(self): return self._act(L_RELU)
    def tanh(self): return self._act(L_TANH)
    def sigmoid(self): return self._act(L_SIGMOID)
    def selu(self): return self._act(L_SELU)
    def leakyrelu(self, a=0.01): return self._act(L_LEAKYRL, a)
    def elu(self, a=1.0): return self._act(L_ELU, a)
    def dropout(self, p): return self._act(L_DROPOUT, p)
    def softmax(self): self._push(L_SOFTMAX, tuple(self.t[-1].shape)); return self

    def _pool(self, fn, k):
        n, h, w, c = self.t[-1].shape
        L = self._push(fn, (n, (h + k - 1) // k, (w + k - 1) // k, c)); L.stride = k; return self

    def maxpool(self, k): return self._pool(L_MAXPOOL, k)
    def avgpool(self, k): return self._pool(L_AVGPOOL, k)
    def minpool(self, k): return self._pool(L_MINPOOL, k)

    def batchnorm(self, m=0.1):
        c = self.t[-1].shape[3]
        L = self._push(L_BATCHNM, tuple(self.t[-1].shape))
        L.w = self.torch.ones(c, device=self.dev); L.b = self._zeros(c); L.dw = self._zeros(c); L.db = self._zeros(c)
        L.aux = self._zeros(tuple(self.t[-2].shape)); L.stat = self._zeros(3 * c); L.xparm = m
        return self

    # ---- forward (Model::forward / _fstep)
    def forward(self, x):
        k, s = self.k, self.s
        assert x.numel() == self.t[0].numel()
        k.call("t4k_copy", _p(x), _p(self.t[0]), x.numel(), s)            # n0 = input
        for i, L in enumerate(self.layers):
            a, y, fn = self.t[i], self.t[i + 1], L.fn
            if fn == L_CONV:
                n, h1, w1, c1 = a.shape; _, h0, w0, c0 = y.shape
                k.call("t4k_conv2d_fwd", _p(a), _p(y), _p(L.w), _p(L.b), n, h1, w1, c1, h0, w0, c0, L.k, L.stride, L.pad, s)
            elif fn == L_LINEAR:
                n = a.shape[0]
                k.call("t4k_linear_fwd", _p(a), _p(L.w), _p(L.b), _p(y), n, y.numel() // n, a.numel() // n, s)
            elif fn == L_FLATTEN:
                k.call("t4k_copy", _p(a), _p(y), a.numel(), s)
            elif fn in (L_RELU, L_TANH, L_SIGMOID, L_SELU, L_LEAKYRL, L_ELU, L_DROPOUT):
                if fn == L_DROPOUT:
                    k.call("t4k_dropout_mask", _p(L.aux), L.aux.numel(), s)
                k.call("t4k_activate", fn, _p(a), _p(y), _p(L.aux), L.xparm, a.numel(), s)
            elif fn == L_SOFTMAX:
                n = a.shape[0]; k.call("t4k_softmax", _p(a), _p(y), n, a.numel() // n, s)
            elif fn in (L_AVGPOOL, L_MAXPOOL, L_MINPOOL):
                n, h1, w1, c = a.shape; _, h0, w0, _ = y.shape
                k.call("t4k_pool", fn, _p(a), _p(y), n, h1, w1, h0, w0, c, L.stride, s)
            elif fn == L_BATCHNM:
                n, h, w, c = a.shape
                k.call("t4k_batchnorm_fwd", _p(a), _p(y), _p(L.aux), _p(L.w), _p(L.b), _p(L.stat), n, h * w, c, s)
            else:
                raise NotImplementedError(fn)
        return self.t[-1]

    def onehot_labels(self, labels_dev):
        out = self.t[-1]; n = out.shape[0]; e = out.numel() // n
        if self.hot is None:
            self.hot = self._zeros((n, 1, e, 1))
        self.k.call("t4k_onehot", _p(labels_dev), _p(self.hot), n, e, self.s)
        self.k.call("t4k_hit", _p(out), _p(self.hot), n, e, _p(self._hit), self.s)
        return self.hot

    def hit(self):
        self.k.call("t4k_sync", self.s)
        return int(self._hit.cpu()[0])

    def loss(self, op, tgt):                                           # Model::loss + Tensor::loss
        k, s = self.k, self.s
        out = self.t[-1]; n = out.shape[0]
        tmp = self.torch.empty_like(out)
        k.call("t4k_copy", _p(out), _p(tmp), out.numel(), s)
        if op == LOSS_MSE:
            k.call("t4k_tt_op", SUB, _p(tmp), _p(tgt), _p(tmp), tmp.numel(), s)
            k.call("t4k_tt_op", MUL, _p(tmp), _p(tmp), _p(tmp), tmp.numel(), s)
            k.call("t4k_reduce", RED_SUM, _p(tmp), tmp.numel(), 0.0, _p(self._scalar), s); sign = 1.0
        elif op == LOSS_BCE:
            k.call("t4k_bce", _p(tgt), _p(tmp), tmp.numel(), _p(self._scalar), s); sign = -1.0
        else:
            if op == LOSS_CE:
                k.call("t4k_math", LN, _p(tmp), 0.0, tmp.numel(), s)
            k.call("t4k_tt_op", MUL, _p(tmp), _p(tgt), _p(tmp), tmp.numel(), s)
            k.call("t4k_reduce", RED_SUM, _p(tmp), tmp.numel(), 0.0, _p(self._scalar), s); sign = -1.0
        k.call("t4k_sync", s)
        return sign * float(self._scalar.cpu()[0]) / n

    # ---- backprop (Model::backprop / _bprep / _bstep)
    def backprop(self, tgt=None):
        k, s = self.k, self.s
        tgt = self.hot if tgt is None else tgt
        out = self.t[-1]
        assert out.numel() == tgt.numel()
        if self.layers[-1].fn in (L_LINEAR, L_SIGMOID, L_SOFTMAX, L_LOGSMAX):
            k.call("t4k_tt_op", SUB, _p(out), _p(tgt), _p(out), out.numel(), s)
        else:
            k.call("t4k_copy", _p(tgt), _p(out), out.numel(), s)
        tr = int(self.train)
        for j, i in enumerate(range(len(self.layers) - 1, -1, -1)):
            L = self.layers[i]; a, y, fn = self.t[i], self.t[i + 1], L.fn
            if fn == L_CONV:
                n, h1, w1, c1 = a.shape; _, h0, w0, c0 = y.shape
                k.call("t4k_conv2d_bwd", _p(a), _p(y), _p(L.aux), _p(L.w), _p(L.dw), _p(L.db),
                       n, h1, w1, c1, h0, w0, c0, L.k, L.stride, L.pad, tr, s)
                k.call("t4k_copy", _p(L.aux), _p(a), a.numel(), s)         # in = dx
            elif fn == L_LINEAR:
                if j == 0:
                    k.call("t4k_copy", _p(y), _p(a), a.numel(), s)
                else:
                    n = a.shape[0]
                    k.call("t4k_linear_bwd", _p(a), _p(L.w), _p(y), _p(a), _p(L.dw), _p(L.db),
                           n, y.numel() // n, a.numel() // n, tr, s)
            elif fn in (L_FLATTEN, L_SIGMOID, L_SOFTMAX, L_LOGSMAX):
                k.call("t4k_copy", _p(y), _p(a), a.numel(), s)
            elif fn in (L_RELU, L_TANH, L_SELU, L_LEAKYRL, L_ELU, L_DROPOUT):
                k.call("t4k_tt_op", MUL, _p(y), _p(L.aux), _p(a), a.numel(), s)
            elif fn in (L_AVGPOOL, L_MAXPOOL, L_MINPOOL):
                n, h1, w1, c = a.shape; _, h0, w0, _ = y.shape
                k.call("t4k_dpool", fn, _p(a), _p(y), n, h1, w1, h0, w0, c, L.stride, s)
            elif fn == L_BATCHNM:
                n, h, w, c = a.shape
                k.call("t4k_batchnorm_bwd", _p(L.w), _p(y), _p(L.aux), _p(a), _p(L.dw), _p(L.db), _p(L.stat), n, h * w, c, tr, s)
            else:
                raise NotImplementedError(fn)
        return self

    # ---- optimizers
    def _params(self):
        for L in self.layers:
            if L.w is not None and L.dw is not None:
                yield L, 0, L.w, L.dw, (L.w.shape[0] if L.fn == L_CONV else 1)
                yield L, 1, L.b, L.db, 1

    def _table(self, need_v):
        recs = b""; n_t = 0; mx = 0
        for L, kk, g, dg, nw in self._params():
            if L.m[kk] is None:
                L.m[kk] = self._zeros(g.shape)
            if need_v and L.m[kk + 2] is None:
                L.m[kk + 2] = self._zeros(g.shape)
            v = L.m[kk + 2] if need_v else L.m[kk]
            recs += struct.pack("<QQQQqii", _p(g), _p(dg), _p(L.m[kk]), _p(v), g.numel(), nw, 0)
            n_t += 1; mx = max(mx, g.numel())
        tab = self.torch.from_numpy(np.frombuffer(recs, np.uint8).copy()).to(self.dev)
        return tab, n_t, mx

    def sgd(self, lr, beta=0.9):
        b = beta if self.iter else 0.0                                   # `_iter ? b : 0`
        self.iter += 1
        if not self.train:
            return self
        if "sgd" not in self._graphs:
            self._graphs["sgd"] = self._table(False)
        tab, n_t, mx = self._graphs["sgd"]
        self.k.call("t4k_opt_multi", 0, _p(tab), n_t, mx, lr, b, 0.0, 0.0, self.s)
        return self

    def adam(self, lr, b1=0.9, b2=0.999):
        self.iter += 1
        if not self.train:
            return self
        if "adam" not in self._graphs:
            self._graphs["adam"] = self._table(True)
        tab, n_t, mx = self._graphs["adam"]
        self.k.call("t4k_opt_multi", 1, _p(tab), n_t, mx, lr, b1, b2, 0.0, self.s)
        return self

    def finalize(self):
        """Re-home every dW/dB into ONE contiguous gradient slab: a single
        all-reduce / a single multi-tensor optimizer launch covers the whole model."""
        offs = []; total = 0
        for L, kk, g, dg, nw in self._params():
            offs.append((L, kk, total, g.numel(), tuple(g.shape))); total += (g.numel() + 63) // 64 * 64
        self.grad_slab = self._zeros(total)
        for L, kk, off, n, shape in offs:
            v = self.grad_slab[off:off + n].view(shape)
            if kk == 0: L.dw = v
            else: L.db = v
        self._graphs.clear()
        return self

    def sync(self):
        self.k.call("t4k_sync", self.s)

    def nparams(self):
        return sum(g.numel() for _, _, g, _, _ in self._params())

    def grads_flat(self):
        return [dg for _, _, _, dg, _ in self._params()]


# ---- the reference's example networks (examples/t4_30e.4th:4-52, README.md:192-215)
def nn_c(m):      # conv10-maxpool2-relu-flatten-linear100-relu-linear10-softmax
    return m.conv2d(10, 0.5).maxpool(2).relu().flatten().linear(100).relu().linear(10).softmax()


def nn_f(m, dropout=True):   # LeNet-style: 2 conv blocks (+dropouts)
    m.conv2d(10, 0.5).maxpool(2).relu().conv2d(20, 0.5)
    if dropout:
        m.dropout(0.5)
    m.maxpool(2).relu().flatten().linear(100)
    if dropout:
        m.dropout(0.5)
    return m.linear(10).softmax()
