"""GPU: dropout draws keyed by the sample's place in the whole batch (include/t4k.h t4k_rand_set_shard).

Every C-ABI entry that draws a dropout mask - stand-alone (`t4k_dropout_mask`) or inside a fused launch (element-wise run,
conv epilogue, split-K fold, folding classifier head) - is run once on the WHOLE batch and once per shard (rank r of 2 sees
rows [r*N/2, (r+1)*N/2) with t4k_rand_set_shard(r, 2) and the same stream position).  The shards' masks must be the rows of
the whole-batch mask bit for bit, and the stream must end at the same position, so that N ranks x batch B train exactly like
one rank x batch N*B with dropout on.  One GPU is enough: the "ranks" run one after the other."""
import ctypes

import numpy as np
import pytest

from test_gpu_parity import Dev, PoolBlock, p, rel, RTOL

pytestmark = pytest.mark.gpu
SEED, OFF = 11, 1 << 16


@pytest.fixture(scope="module")
def dev(t4k):
    return Dev(t4k)


def _both(t4k, run, N):
    """run(lo, hi) -> dict of arrays with a leading sample axis; returns (whole, concatenated shards, offsets)"""
    t4k.call("t4k_rand_set_shard", 0, 1)
    t4k.call("t4k_rand_init", SEED); t4k.call("t4k_rand_set_offset", OFF)
    whole = run(0, N); off_whole = t4k.lib.t4k_rand_offset()
    parts = []; offs = []
    try:
        for r in range(2):
            t4k.call("t4k_rand_set_shard", r, 2)
            t4k.call("t4k_rand_init", SEED); t4k.call("t4k_rand_set_offset", OFF)
            parts.append(run(r * N // 2, (r + 1) * N // 2)); offs.append(t4k.lib.t4k_rand_offset())
    finally:
        t4k.call("t4k_rand_set_shard", 0, 1)
    cat = {k: np.concatenate([q[k] for q in parts], axis=0) for k in whole}
    assert offs[0] == offs[1] == off_whole, "the stream must move by the whole batch's draw on every rank"
    return whole, cat


def test_dropout_mask_entry(t4k, dev):
    N, E = 16, 100
    def run(lo, hi):
        m = dev.zeros(((hi - lo), E)); t4k.call("t4k_dropout_mask", p(m), (hi - lo) * E, None); return {"mask": dev.down(m)}
    whole, cat = _both(t4k, run, N)
    assert np.array_equal(whole["mask"], cat["mask"])
    t4k.call("t4k_rand_init", SEED); t4k.call("t4k_rand_set_offset", OFF)      # and without a shard it IS t4k_rand(uniform)
    r = dev.zeros((N, E)); t4k.call("t4k_rand", p(r), N * E, 0, 0.0, 1.0, None)
    assert np.array_equal(dev.down(r), whole["mask"])


@pytest.mark.parametrize("N,E0,E1", [(128, 100, 980), (64, 16, 40), (256, 128, 1024)])   # split-K fold epilogue / small head / plain GEMM + launch
def test_linear_dropout_epilogue(t4k, dev, oracle, N, E0, E1):
    rng = np.random.default_rng(E0)
    X = rng.standard_normal((N, E1)).astype(np.float32); W = (rng.standard_normal((E0, E1)) * 0.1).astype(np.float32)
    b = rng.standard_normal(E0).astype(np.float32)
    dW, dB = dev.up(W), dev.up(b)
    def run(lo, hi):
        n = hi - lo
        dY, dF, dA = dev.zeros((n, E0)), dev.zeros((n, E0)), dev.zeros((n, E0))
        t4k.call("t4k_linear_act_fwd", p(dev.up(X[lo:hi])), p(dW), p(dB), p(dY), oracle.L_DROPOUT, 0.5, p(dF), p(dA), n, E0, E1, None)
        return {"mask": dev.down(dF), "act": dev.down(dA)}
    whole, cat = _both(t4k, run, N)
    assert np.array_equal(whole["mask"], cat["mask"])
    assert rel(cat["act"], whole["act"]) < RTOL


def test_folding_classifier_head(t4k, dev, oracle):
    N, E1, H, E2 = 128, 980, 100, 10                     # LeNet head: the 100 -> 10 launch folds the 980 -> 100 split-K slabs and draws the mask
    rng = np.random.default_rng(4)
    X = rng.standard_normal((N, E1)).astype(np.float32)
    W1 = (rng.standard_normal((H, E1)) * 0.05).astype(np.float32); b1 = rng.standard_normal(H).astype(np.float32)
    W2 = (rng.standard_normal((E2, H)) * 0.2).astype(np.float32); b2 = rng.standard_normal(E2).astype(np.float32)
    dW1, dB1, dW2, dB2 = dev.up(W1), dev.up(b1), dev.up(W2), dev.up(b2)
    def run(lo, hi):
        n = hi - lo
        dY1, dF, dA1, dY2, dP2 = dev.zeros((n, H)), dev.zeros((n, H)), dev.zeros((n, H)), dev.zeros((n, E2)), dev.zeros((n, E2))
        t4k.call("t4k_mlp_head_fwd", p(dev.up(X[lo:hi])), p(dW1), p(dB1), p(dY1), oracle.L_DROPOUT, 0.5, p(dF), p(dA1), p(dW2), p(dB2), p(dY2), p(dP2), n, H, E1, E2, None)
        return {"mask": dev.down(dF), "prob": dev.down(dP2)}
    whole, cat = _both(t4k, run, N)
    assert np.array_equal(whole["mask"], cat["mask"])
    assert rel(cat["prob"], whole["prob"]) < RTOL


@pytest.mark.parametrize("C", [5, 8])
def test_elementwise_run_with_dropout(t4k, dev, oracle, C):
    N, H1 = 6, 12
    X = np.random.default_rng(C).standard_normal((N, H1, H1, C)).astype(np.float32)
    def run(lo, hi):
        n = hi - lo
        d = {"pre_mask": dev.zeros((n, H1, H1, C)), "pre_out": dev.zeros((n, H1, H1, C)), "pool_out": dev.zeros((n, H1 // 2, H1 // 2, C))}
        blk = PoolBlock(); blk.KS = 2
        blk.pre_layer, blk.pre_alpha = oracle.L_DROPOUT, 0.5; blk.pre_mask = p(d["pre_mask"]); blk.pre_out = p(d["pre_out"])
        blk.pool_layer = oracle.L_MAXPOOL; blk.pool_out = p(d["pool_out"])
        t4k.call("t4k_poolblock_fwd", p(dev.up(X[lo:hi])), ctypes.byref(blk), n, H1, H1, H1 // 2, H1 // 2, C, None)
        return {k: dev.down(v) for k, v in d.items()}
    whole, cat = _both(t4k, run, N)
    for k in whole:
        assert np.array_equal(whole[k], cat[k]), k


@pytest.mark.parametrize("N,H,C1,C0", [(8, 14, 10, 20), (4, 8, 6, 40), (4, 6, 5, 8)])      # LeNet conv2 block; two channel tiles; narrow
def test_conv_epilogue_dropout(t4k, dev, oracle, N, H, C1, C0):
    rng = np.random.default_rng(C0)
    X = rng.standard_normal((N, H, H, C1)).astype(np.float32); F = (rng.standard_normal((C1, 3, 3, C0)) * 0.3).astype(np.float32)
    B = rng.standard_normal(C0).astype(np.float32)
    dF, dB = dev.up(F), dev.up(B)
    def run(lo, hi):
        n = hi - lo
        d = {"pre_mask": dev.zeros((n, H, H, C0)), "pre_out": dev.zeros((n, H, H, C0)), "pool_out": dev.zeros((n, H // 2, H // 2, C0))}
        dY = dev.zeros((n, H, H, C0))
        blk = PoolBlock(); blk.KS = 2; blk.pool_layer = oracle.L_MAXPOOL; blk.pool_out = p(d["pool_out"])
        blk.pre_layer, blk.pre_alpha = oracle.L_DROPOUT, 0.5; blk.pre_mask = p(d["pre_mask"]); blk.pre_out = p(d["pre_out"])
        t4k.call("t4k_conv2d_block_fwd", p(dev.up(X[lo:hi])), None, p(dY), p(dF), p(dB), ctypes.byref(blk), n, H, H, C1, H, H, C0, 3, 1, 1, None)
        out = {k: dev.down(v) for k, v in d.items()}; out["conv"] = dev.down(dY)
        return out
    whole, cat = _both(t4k, run, N)
    assert np.array_equal(whole["pre_mask"], cat["pre_mask"])
    assert np.array_equal(whole["conv"], cat["conv"])        # per-sample work: bitwise the same whatever the batch
    assert np.array_equal(whole["pool_out"], cat["pool_out"])
