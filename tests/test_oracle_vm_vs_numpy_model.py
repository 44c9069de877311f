"""CPU: the oracle-backed VM against an independent restatement of the host orchestration.

`oracle/ten4_oracle` links the SAME host sources (tensorforth_amd/host/*.cpp) as the product VM, so the golden files it writes
cannot catch an orchestration bug both VMs share.  `oracle/t4oracle.py`'s OracleModel is a separate, numpy
restatement of nn::Model (layer factory, forward / backprop order, in-place gradient convention, loss, SGD / momentum / Adam) that
shares only the per-kernel oracle functions.  Here the two are run on the same seed and the same Forth-level program:
every number the scripts print (probabilities, loss, bias / weight gradients, dX, weights after SGD and Adam, dropout-mask sums)
must agree."""
import os

import numpy as np
import pytest

from vm_util import SCRIPTS, TEN4, TEN4_ORACLE, ROOT, numbers_after, run_vm


@pytest.fixture(scope="module")
def oracle_vm():
    if not os.path.exists(TEN4_ORACLE):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ten4_oracle"], check=True, capture_output=True)
    return TEN4_ORACLE


def _urand(oracle, shape):
    a = np.zeros(shape, np.float32)
    oracle.lib().t4o_rand(oracle.P(a), a.size, 0, 0.0, 1.0)
    return a


def _nrand(oracle, shape):
    a = np.zeros(shape, np.float32)
    oracle.lib().t4o_rand(oracle.P(a), a.size, 1, 0.0, 1.0)
    return a


def _close(got, want, tol=2.5e-4, rtol=2e-4):
    got = np.asarray(got, np.float64).ravel(); want = np.asarray(want, np.float64).ravel()
    assert got.shape == want.shape
    assert np.all(np.abs(got - want) <= tol + rtol * np.abs(want)), (got[:6], want[:6])


def test_cnn_step_script_equals_numpy_model(oracle_vm, oracle):
    _cnn_step_vs_numpy(oracle_vm, oracle)


def _cnn_step_vs_numpy(binary, oracle, env=None):
    out = run_vm(binary, os.path.join(SCRIPTS, "cnn_step.4th"), seed=1, env_extra=env)
    m = oracle.OracleModel(4, 28, 28, 1, seed=1)
    m.conv2d(10, 0.5).maxpool(2).relu().conv2d(20, 0.5).maxpool(2).relu().flatten().linear(100).linear(10).softmax()
    img = _urand(oracle, (4, 28, 28, 1))
    hot = np.zeros((4, 1, 10, 1), np.float32)
    for flat in (3, 15, 20, 39):
        hot.ravel()[flat] = 1.0
    _close(numbers_after(out, "probs", 40), m.forward(img))
    _close(numbers_after(out, "ce", 1), [m.loss(oracle.LOSS_CE, hot)])
    m.backprop(hot)
    L = m.layers
    _close(numbers_after(out, "g_conv0_b", 10), L[0].db)
    _close(numbers_after(out, "g_lin8_b", 10), L[8].db)
    _close(numbers_after(out, "g_conv0_w", 1), [L[0].dw.sum(dtype=np.float64)], tol=2e-3)
    _close(numbers_after(out, "g_conv3_w", 1), [L[3].dw.sum(dtype=np.float64)], tol=2e-3)
    _close(numbers_after(out, "g_lin7_w", 1), [L[7].dw.sum(dtype=np.float64)], tol=2e-2)
    _close(numbers_after(out, "dx_in", 1), [m.t[0].sum(dtype=np.float64)], tol=2e-3)
    m.sgd(0.01, 0.0)
    _close(numbers_after(out, "w0", 90), L[0].w)
    _close(numbers_after(out, "b8", 10), L[8].b)
    m.forward(img); m.backprop(hot); m.sgd(0.01, 0.0)
    _close(numbers_after(out, "probs2", 40), m.forward(img), tol=1e-3)
    # the script's second model (Adam from its first step on) is built later in the SAME random stream
    import ctypes
    lib = oracle.lib(); lib.t4o_rand_offset.restype = ctypes.c_uint64; lib.t4o_rand_set_offset.argtypes = [ctypes.c_uint64]
    off = lib.t4o_rand_offset()
    m2 = oracle.OracleModel(4, 28, 28, 1, seed=1); lib.t4o_rand_set_offset(off)
    m2.conv2d(10, 0.5).maxpool(2).relu().flatten().linear(10).softmax()
    m2.forward(img); m2.backprop(hot); m2.adam(0.001)
    _close(numbers_after(out, "w0a", 90), m2.layers[0].w)
    m2.forward(img); m2.backprop(hot); m2.adam(0.001)
    _close(numbers_after(out, "probs3", 40), m2.forward(img), tol=1e-3)


def test_train_loop_script_with_dropout_momentum_and_adam_equals_numpy_model(oracle_vm, oracle):
    _train_loop_vs_numpy(oracle_vm, oracle)


def _train_loop_vs_numpy(binary, oracle, env=None):
    out = run_vm(binary, os.path.join(SCRIPTS, "cnn_train_loop.4th"), seed=1, env_extra=env)
    m = oracle.OracleModel(8, 28, 28, 1, seed=1)
    m.conv2d(10, 0.5).maxpool(2).relu().conv2d(20, 0.5).dropout(0.5).maxpool(2).relu().flatten().linear(100).dropout(0.5).linear(10).softmax()
    img = _urand(oracle, (8, 28, 28, 1))
    hot = np.zeros((8, 1, 10, 1), np.float32)
    for i in range(8):
        hot.ravel()[i * 10 + (i * 3) % 10] = 1.0
    losses = []
    for _ in range(6):
        m.forward(img); losses.append(m.loss(oracle.LOSS_CE, hot)); m.backprop(hot); m.sgd(0.002, 0.9)
    toks = out.split()
    i0 = toks.index("mask1")
    # the six step losses are the six numbers printed right before the first label
    got = [float(t) for t in toks[:i0] if t.replace(".", "", 1).replace("-", "", 1).replace("e", "", 1).replace("+", "", 1).isdigit() and "." in t][-6:]
    _close(got, losses, tol=1e-3, rtol=1e-3)
    L = m.layers
    _close(numbers_after(out, "mask1", 1), [L[4].aux.sum(dtype=np.float64)], tol=0.5)
    _close(numbers_after(out, "mask2", 1), [L[9].aux.sum(dtype=np.float64)], tol=0.5)
    for label, arr in (("w0", L[0].w), ("w3", L[3].w), ("w8", L[8].w), ("w10", L[10].w)):
        _close(numbers_after(out, label, 1), [arr.sum(dtype=np.float64)], tol=5e-3, rtol=1e-3)
    _close(numbers_after(out, "b10", 10), L[10].b, tol=1e-3)
    # second model of the script: tanh + dropout + sigmoid MLP, MSE, Adam
    mlp = oracle.OracleModel.__new__(oracle.OracleModel)          # same Philox stream continues: do not re-seed
    mlp.t = [np.zeros((8, 1, 16, 1), np.float32)]; mlp.layers = []; mlp.train = True; mlp.iter = 0; mlp.epoch = 0; mlp.hot = None; mlp.hit = 0
    m.forward(img)                                                 # the script's `img forward` before building the MLP draws two masks
    mlp.linear(12).tanh().dropout(0.2).linear(4).sigmoid()
    x = _nrand(oracle, (8, 1, 16, 1)); y = _urand(oracle, (8, 1, 4, 1))
    for _ in range(5):                                             # `4 for ... next` runs 5 times
        mlp.forward(x); mlp.loss(oracle.LOSS_MSE, y); mlp.backprop(y); mlp.adam(0.01)
    _close(numbers_after(out, "aw0", 1), [mlp.layers[0].w.sum(dtype=np.float64)], tol=5e-3, rtol=1e-3)
    w3 = mlp.layers[3].w.reshape(4, 12)                            # printed with the width elision of the reference's printer (first 3 ... last 3)
    _close(numbers_after(out, "aw3", 24), w3[:, [0, 1, 2, 9, 10, 11]], tol=1e-3)


# every launch plan of the PRODUCT host against the numpy model - not against goldens the same host/model.cpp wrote: the pattern matcher
# (stack_at, head / run fusion, fold inside the optimizer, lazy first-layer dX) decides which kernels run, the numpy model knows none of it
PLANS = [{}, {"T4_OPT_FOLD": "0"}, {"T4_LAZY_DX0": "0"}, {"T4_HEAD_BWD": "0"}, {"T4_STACK_HEAD": "0"}, {"T4_STACK_HEAD": "0", "T4_HEAD_BWD": "0"}, {"T4_STACK": "0"}, {"T4_FUSE": "0"},
         {"T4_OPT_FOLD": "0", "T4_LAZY_DX0": "0", "T4_HEAD_BWD": "0", "T4_STACK_HEAD": "0"}]
PLAN_IDS = ["default", "fold-apart", "eager-dx0", "head-bwd-apart", "head-fwd-apart", "head-apart", "no-stack", "one-launch-per-layer", "round2-plan"]


@pytest.mark.gpu
@pytest.mark.parametrize("env", PLANS, ids=PLAN_IDS)
def test_every_lenet_launch_plan_of_the_product_vm_equals_the_numpy_model(oracle, env):
    _cnn_step_vs_numpy(TEN4, oracle, env)                     # forward, CE, every gradient, dX of the input (`0 n@` after backprop: produced on demand under the default plan), SGD, Adam
    _train_loop_vs_numpy(TEN4, oracle, env)                   # both dropouts, momentum SGD over six steps, a second model (tanh / dropout / sigmoid MLP, MSE, Adam) on the same stream
