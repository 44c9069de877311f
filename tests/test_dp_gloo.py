"""Data-parallel path on CPU: world_size 2 over gloo (127.0.0.1).

Each rank builds the same replica (oracle model, same Philox seed), runs forward/backprop on ITS rows of
the global batch, all-reduces the flattened gradient slab with tensorforth_amd.dp (the code bench.py
uses on the GPUs), applies SGD, and the result must equal a single process training on the whole batch:
gradients are raw batch sums in the reference (gradient.cu:63-126), so SUM over shards is exact up to
fp32 summation order.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GLOBAL_N = 8


def _build(n, dropout=False):
    import t4oracle
    m = t4oracle.OracleModel(n, 12, 12, 1, seed=99)
    if dropout:      # dropout behind the conv AND inside the head, as in the LeNet net of bench.py (t4_30e nn_f)
        m.conv2d(4, 0.5).dropout(0.5).maxpool(2).relu().flatten().linear(16).dropout(0.25).linear(10).softmax()
    else:
        m.conv2d(4, 0.5).maxpool(2).relu().flatten().linear(16).relu().linear(10).softmax()
    return m


def _data():
    rng = np.random.default_rng(5)
    x = rng.random((GLOBAL_N, 12, 12, 1)).astype(np.float32)
    lab = rng.integers(0, 10, GLOBAL_N).astype(np.uint32)
    return x, lab


def _grads(m):
    return [g for L in m.layers for g in (L.dw, L.db) if g is not None]


def _params(m):
    return [g for L in m.layers for g in (L.w, L.b) if g is not None]


def _masks(m):
    import t4oracle
    return [L.aux for L in m.layers if L.fn == t4oracle.L_DROPOUT]


def _worker(rank, world, port, out_dir, dropout=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tensorforth_amd import dp
    x, lab = _data()
    lo, hi = dp.shard_rows(GLOBAL_N, rank, world)
    m = _build(hi - lo, dropout)
    if dropout:                                             # weights above were replicated draws; masks are keyed by sample from here on
        import t4oracle
        assert t4oracle.lib().t4o_rand_set_shard(rank, world) == 0
    for it in range(2):
        m.forward(x[lo:hi]); m.onehot_labels(lab[lo:hi]); m.backprop()
        gs = _grads(m)
        slab = torch.from_numpy(np.concatenate([g.ravel() for g in gs]))          # the gradient slab
        if it == 0:
            dp.allreduce_grad_slab(slab)                                            # one collective after backprop
        else:                                                                       # overlapped: tail first (as the VM's hook reports it), head last
            red = dp.OverlappedSlabReducer(slab, None)
            offs = np.cumsum([0] + [g.size for g in gs])
            for k in range(len(gs) - 2, -1, -2):                                    # layers complete in reverse order: (dW, dB) pairs
                red.on_layer(k // 2, int(offs[k]), int(offs[k + 2] - offs[k]))
            assert red.cut is not None and 0 < red.cut < slab.numel()
            red.finish()
        off = 0
        for g in gs:
            g[...] = slab[off:off + g.size].numpy().reshape(g.shape); off += g.size
        m.sgd(0.05, 0.0)
    loss_n = float(m.loss_ce()) * (hi - lo) if hasattr(m, "loss_ce") else 0.0
    tot = dp.allreduce_scalars([loss_n, float(hi - lo)])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), *[p for p in _params(m)], tot=np.array(tot), **{"mask%d" % i: a for i, a in enumerate(_masks(m))})
    dist.barrier()
    dist.destroy_process_group()


def test_shard_rows_and_offsets():
    from tensorforth_amd import dp
    assert [dp.shard_rows(1024, r, 8) for r in (0, 7)] == [(0, 128), (896, 1024)]
    with pytest.raises(ValueError):
        dp.shard_rows(10, 0, 4)
    offs = {dp.rank_rng_offset(r) for r in range(8)}
    assert len(offs) == 8 and min(offs) >= 1 << 36 and all(o % 4 == 0 for o in offs)


def test_two_rank_gloo_equals_single_process_large_batch(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 400)
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    # single process, whole batch
    x, lab = _data()
    m = _build(GLOBAL_N)
    for _ in range(2):
        m.forward(x); m.onehot_labels(lab); m.backprop(); m.sgd(0.05, 0.0)
    ref = _params(m)
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npz")); r1 = np.load(os.path.join(str(tmp_path), "rank1.npz"))
    for i, p in enumerate(ref):
        a, b = r0["arr_%d" % i], r1["arr_%d" % i]
        assert np.array_equal(a, b), "replicas diverged"                            # same slab on both ranks => bit-identical updates
        np.testing.assert_allclose(a, p, rtol=2e-5, atol=2e-6)                       # == large-batch update up to summation order
    assert r0["tot"][1] == GLOBAL_N


def test_two_rank_gloo_with_dropout_equals_single_process_large_batch(tmp_path):
    """dropout draws are keyed by the sample's place in the WHOLE batch (t4k_rand_set_shard / t4o_rand_set_shard), so
    2 ranks x 4 samples draw exactly the masks of 1 rank x 8 samples and the two runs train to the same weights."""
    world = 2
    port = 29900 + (os.getpid() % 90)
    mp.start_processes(_worker, args=(world, port, str(tmp_path), True), nprocs=world, join=True, start_method="spawn")
    x, lab = _data()
    m = _build(GLOBAL_N, dropout=True)
    for _ in range(2):
        m.forward(x); m.onehot_labels(lab); m.backprop(); m.sgd(0.05, 0.0)
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(world)]
    for i, mk in enumerate(_masks(m)):                      # derivative masks of the last step (0/1): bit-exact rows of the big-batch mask
        got = np.concatenate([r[k]["mask%d" % i] for k in range(world)], axis=0)
        assert np.array_equal(got.reshape(mk.shape), mk), "dropout mask %d differs from the whole-batch draw" % i
    for i, p in enumerate(_params(m)):
        assert np.array_equal(r[0]["arr_%d" % i], r[1]["arr_%d" % i]), "replicas diverged"
        np.testing.assert_allclose(r[0]["arr_%d" % i], p, rtol=2e-5, atol=2e-6)


def test_sharded_batchnorm_sums_reproduce_whole_batch_oracle():
    """The data-parallel batch norm of reduce.hip (k_bn_sums / k_bn_fin_sync): per-rank column sums, SUM over ranks,
    finalise over world x NHW; dgamma/dbeta take the LOCAL sums over the GLOBAL count so that the slab all-reduce that
    follows yields the whole-batch means.  Restated in numpy per shard and checked against the oracle on the whole batch."""
    import t4oracle
    o = t4oracle.lib(); P = t4oracle.P
    rng = np.random.default_rng(3)
    N, HW, C, world = 8, 36, 5, 2
    x = (rng.standard_normal((N, HW, C)) * 2 + 1).astype(np.float32)
    g = rng.standard_normal(C).astype(np.float32); b = rng.standard_normal(C).astype(np.float32)
    gy = rng.standard_normal(x.shape).astype(np.float32)
    y = np.zeros_like(x); xh = np.zeros_like(x); stat = np.zeros(3 * C, np.float32)
    o.t4o_batchnorm_fwd(P(x), P(y), P(xh), P(g), P(b), P(stat), N, HW, C)
    fstat = stat.copy()                                                       # backward reuses the stat rows
    DX = np.zeros_like(x); DW = np.zeros(C, np.float32); DB = np.zeros(C, np.float32)
    o.t4o_batchnorm_bwd(P(g), P(gy), P(xh), P(DX), P(DW), P(DB), P(stat), N, HW, C, 1)
    from tensorforth_amd import dp
    nhw_g = np.float32(N * HW)
    shards = [slice(*dp.shard_rows(N, r, world)) for r in range(world)]
    sums = sum(np.stack([x[s].reshape(-1, C).sum(0), (x[s].reshape(-1, C) ** 2).sum(0)]) for s in shards)   # the all-reduce
    avg = sums[0] / nhw_g; istd = 1.0 / (np.sqrt(np.maximum(sums[1] / nhw_g - avg * avg, 0)) + np.float32(1e-6))
    np.testing.assert_allclose(avg, fstat[C:2 * C], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(istd, fstat[:C], rtol=1e-4)
    slab = np.zeros((2, C), np.float32)
    for s in shards:                                                          # each rank's dbeta/dgamma, then the slab SUM
        loc = np.stack([gy[s].reshape(-1, C).sum(0), (gy[s] * xh[s]).reshape(-1, C).sum(0)])
        slab += loc / nhw_g
    np.testing.assert_allclose(slab[0], DB, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(slab[1], DW, rtol=1e-4, atol=1e-6)
