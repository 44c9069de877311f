"""Pin the CPU oracle against the reference's own known-answer scripts
(tests/golden/kat_reference_examples.json, transcribed from examples/t4_30a/b/c, t4_20a,
t4_22a expected-value comments).  CPU only."""
import ctypes
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat_reference_examples.json")))
TOL = 6e-5      # printed with 4 decimals


def close(a, b, tol=TOL):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.max(np.abs(a - b)) <= tol, (a, b)


def test_t4_30a_linear_forward(oracle):
    k = KAT["t4_30a"]
    m = oracle.OracleModel(*k["model_in"]).linear(3)
    m.layers[0].w[...] = np.array(k["w0"], np.float32); m.layers[0].b[...] = np.array(k["b0"], np.float32)
    out = m.forward(np.array(k["input"], np.float32))
    close(out, k["expect_out"], 1e-5)


def _mlp(oracle, k, hidden):
    m = oracle.OracleModel(*k["model_in"]).linear(hidden).sigmoid().linear(2).sigmoid()
    m.layers[0].w[...] = np.array(k["w0"], np.float32); m.layers[0].b[...] = np.array(k["b0"], np.float32)
    m.layers[2].w[...] = np.array(k["w2"], np.float32); m.layers[2].b[...] = np.array(k["b2"], np.float32)
    return m


def test_t4_30b_mazur_backprop(oracle):
    k = KAT["t4_30b"]; e = k["expect"]
    m = _mlp(oracle, k, 3)
    m.forward(np.array(k["input"], np.float32))
    close(m.t[1], e["L1_in"]); close(m.layers[1].aux, e["L1_mask"]); close(m.t[2], e["L2_in"])
    close(m.t[3], e["L3_in"]); close(m.layers[3].aux, e["L3_mask"]); close(m.t[4], e["out"])
    tgt = np.array(k["target"], np.float32)
    assert abs(m.loss(oracle.LOSS_MSE, tgt) - e["loss_mse"]) < 2e-6
    m.backprop(tgt)
    close(m.t[4], e["L4_dY"]); close(m.t[3], e["L3_dX"])          # sigmoid pass-through (quirk a-14)
    close(m.layers[2].db, e["L2_dB"]); close(m.layers[2].dw, e["L2_dW"]); close(m.t[2], e["L2_dX"])
    close(m.t[1], e["L1_dX"]); close(m.layers[0].db, e["L0_dB"]); close(m.layers[0].dw, e["L0_dW"])
    close(m.t[0], e["L0_dX"])
    m.sgd(e["sgd"]["lr"], e["sgd"]["beta"])
    close(m.layers[2].w, e["L2_W_after"]); close(m.layers[2].b, e["L2_B_after"])
    close(m.layers[0].w, e["L0_W_after"]); close(m.layers[0].b, e["L0_B_after"])
    assert not m.layers[2].dw.any() and not m.layers[0].db.any()    # zeroed after the update


def test_t4_30c_batch_sum_semantics(oracle):
    k = KAT["t4_30c"]; e = k["expect"]
    m = _mlp(oracle, k, 2)
    m.forward(np.array(k["input"], np.float32))
    for n in range(3):
        close(m.t[1][n], e["L1_in"]); close(m.t[2][n], e["L2_in"]); close(m.t[3][n], e["L3_in"]); close(m.t[4][n], e["out"])
    tgt = np.array(k["target"], np.float32)
    assert abs(m.loss(oracle.LOSS_MSE, tgt) - e["loss_mse"]) < 2e-6
    m.backprop(tgt)
    for n in range(3):
        # L4_dY / L1_dX come from free-hand comments (t4_30c.4th:43,48: -0.2172 is a typo for
        # -0.2171 = 0.7729-0.99; 3x it gives the listed dB -0.6512); the `verify` lines are strict
        close(m.t[4][n], e["L4_dY"], 2e-4); close(m.t[1][n], e["L1_dX"], 2e-4); close(m.t[0][n], e["L0_dX"])
    close(m.layers[2].db, e["L2_dB"], 1.1e-4)                      # sums of 3 rounded terms
    dw2 = m.layers[2].dw.ravel(); ex2 = np.array(e["L2_dW"]).ravel()
    close(dw2[[0, 1, 3]], ex2[[0, 1, 3]], 1.1e-4)                  # [1][0] is listed as -0.3836 (free-hand
    assert abs(dw2[2] - (-0.3863)) < 1.1e-4                        # digit swap of -0.3863 = 3*(-0.2171*0.5933))
    close(m.layers[0].db, e["L0_dB"], 1.1e-4); close(m.layers[0].dw, e["L0_dW"], 1.1e-4)
    m.sgd(e["sgd"]["lr"], e["sgd"]["beta"])
    close(m.layers[2].b, e["L2_B_after"], 1.1e-4)
    close(m.layers[0].w, e["L0_W_after"], 1.1e-4); close(m.layers[0].b, e["L0_B_after"], 1.1e-4)


def test_t4_20a_matrix_words(oracle):
    k = KAT["t4_20a"]
    A = np.array(k["matmul"]["A"], np.float32); ones = np.ones(k["matmul"]["B_ones"], np.float32)
    assert np.array_equal(oracle.gemm(A, ones), np.array(k["matmul"]["expect"], np.float32))
    o = oracle.lib()
    s = np.zeros_like(A); o.t4o_tt_op(oracle.ADD, oracle.P(A), oracle.P(np.ones_like(A)), oracle.P(s), A.size)
    assert np.array_equal(s, np.array(k["add"]["expect"], np.float32))
    d = np.zeros_like(A); o.t4o_tt_op(oracle.SUB, oracle.P(A), oracle.P(np.ones_like(A)), oracle.P(d), A.size)
    assert np.array_equal(d, np.array(k["sub"]["expect"], np.float32))
    A2 = np.array(k["matmul2"]["A"], np.float32)
    c = oracle.gemm(A2, ones)
    assert np.array_equal(c, np.array(k["matmul2"]["expect"], np.float32))
    h = np.ones((2, 2), np.float32); o.t4o_ts_op(oracle.MUL, oracle.P(h), 0.5, oracle.P(h), 4)
    o.t4o_tt_op(oracle.MUL, oracle.P(c), oracle.P(h), oracle.P(c), 4)
    assert np.array_equal(c, np.array(k["hadamard"]["expect"], np.float32))


def _det(oracle, A):
    o = oracle.lib(); K = A.shape[0]
    lu = A.copy(); piv = np.zeros(K, np.int32); st = ctypes.c_int(0)
    o.t4o_plu(oracle.P(lu), None, oracle.P(piv), K, ctypes.byref(st))
    ld = np.zeros(1, np.float32); sg = ctypes.c_int(0)
    o.t4o_logdet(oracle.P(lu), K, oracle.P(ld), ctypes.byref(sg))
    swaps = int(np.sum(piv != np.arange(K)))
    return math_exp(ld[0]) * (1 if swaps % 2 == 0 else -1) * sg.value      # Tensor::det tensor.cu:431-456


def math_exp(x):
    return float(np.exp(np.float32(x)))


def test_t4_22a_linear_algebra(oracle):
    k = KAT["t4_22a"]; o = oracle.lib()
    A = np.array(k["A"], np.float32)
    assert abs(_det(oracle, A) - k["det"]) < 1e-4
    for fn in ("gj", "lu"):
        a = A.copy(); I = np.eye(3, dtype=np.float32); st = ctypes.c_int(0); piv = np.zeros(3, np.int32)
        if fn == "gj":
            o.t4o_inverse(oracle.P(a), oracle.P(I), 3, ctypes.byref(st))
        else:
            o.t4o_lu_inverse(oracle.P(a), oracle.P(I), oracle.P(piv), 3, ctypes.byref(st))
        assert st.value == 0
        assert np.allclose(A @ I, np.eye(3), atol=1e-5)
    # PLU reconstruction: P @ L @ U == A  (t4_22a.4th:20-41)
    B = np.array(k["A_plu"], np.float32)
    lu = B.copy(); Pm = np.eye(3, dtype=np.float32); piv = np.zeros(3, np.int32); st = ctypes.c_int(0)
    o.t4o_plu(oracle.P(lu), oracle.P(Pm), oracle.P(piv), 3, ctypes.byref(st))
    Lm = lu.copy(); o.t4o_lu_extract(oracle.P(Lm), 0, 3)
    Um = lu.copy(); o.t4o_lu_extract(oracle.P(Um), 1, 3)
    assert np.allclose(Pm @ Lm @ Um, B, atol=1e-5)
    # solve B = A X via luinv (tenvm.cpp:369-384)
    s = k["solve"]; As = np.array(s["A"], np.float32); I = np.eye(3, dtype=np.float32)
    o.t4o_lu_inverse(oracle.P(As.copy()), oracle.P(I), oracle.P(piv), 3, ctypes.byref(st))
    X = oracle.gemm(I, np.array(s["B"], np.float32).reshape(3, 1))
    assert np.allclose(X.ravel(), s["X"], atol=2e-4)


def test_singular_matrix_reports_column(oracle):
    o = oracle.lib()
    a = np.array([[1, 2], [2, 4]], np.float32); I = np.eye(2, dtype=np.float32); st = ctypes.c_int(0)
    o.t4o_inverse(oracle.P(a), oracle.P(I), 2, ctypes.byref(st))
    assert st.value == 2
