"""GPU: the non-default dispatch paths of the GEMM / linear kernels cannot rot, and the gated one-launch kernels survive a busy device.

* the integer-exact products run on the default dispatch, with the gated kernels switched off, and (LAB library, opt-in) under every
  `T4K_GEMM_*` / head LAB switch (csrc/gemm.hip, linear_small.hip), each in a process of its own: plain / transposed / alpha-beta GEMMs incl. ragged and sliver shapes, and the linear layer both ways (forward with bias,
  backward dW += dY^T X, dB += column sums, dX = dY W written IN PLACE over X - the arrival-gate path) - entries in {-2..2} keep every
  fp32 sum exact, so whatever kernel the switch selects must reproduce numpy's integer result bit for bit;
* the same set runs while a bandwidth-hogging elementwise kernel chain occupies the device on another stream (the situation of a
  concurrent RCCL kernel, T4_DP_OVERLAP): results unchanged and no bounded-wait timeout reported by t4k_sync (csrc/t4k_common.h)."""
import os
import subprocess
import sys

import pytest

from vm_util import ROOT

pytestmark = pytest.mark.gpu

_SCRIPT = r'''
import ctypes, os, sys
sys.path.insert(0, os.environ["T4_ROOT"])
import numpy as np, torch
from tensorforth_amd.lib import load
k = load(os.environ.get("T4K_LIB") or None); k.init(0)
if os.environ.get("GATES_OFF") == "1": k.lib.t4k_gates_enable(0)
k.call("t4k_set_default_stream", None)
p = lambda t: t.data_ptr()
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rng = np.random.default_rng(11)
hog = os.environ.get("HOG") == "1"
side = torch.cuda.Stream() if hog else None
big = torch.zeros(512 << 20 >> 2, device="cuda") if hog else None       # 512 MB

def busy():
    if hog:
        with torch.cuda.stream(side):
            for _ in range(6):
                big.mul_(1.0001).add_(1.0)

def ints(*shape): return rng.integers(-2, 3, shape).astype(np.float32)
fails = []
for (M, N, K, tA, tB, al, be) in [(1024, 1024, 1024, 0, 0, 1.0, 0.0), (1024, 1024, 784, 0, 1, 1.0, 0.0), (512, 1024, 1024, 0, 1, 2.0, -1.0), (1000, 1028, 256, 0, 1, 2.0, -1.0),
                                  (996, 1000, 384, 1, 0, 1.0, 1.0), (40, 72, 64, 0, 1, 1.0, 0.0), (200, 100, 832, 0, 1, 2.0, -1.0), (256, 512, 784, 0, 1, 1.0, 0.0),
                                  (2048, 2048, 256, 0, 0, 1.0, 0.0), (2048, 2048, 328, 0, 1, 2.0, -1.0), (1024, 512, 256, 1, 1, 1.0, 0.0), (320, 1088, 384, 1, 1, 1.0, 0.0), (2048, 1024, 512, 1, 1, 1.0, 0.0), (128, 100, 980, 0, 1, 1.0, 0.0),
                                  (1024, 1024, 784, 0, 0, 1.0, 0.0), (1024, 1024, 1004, 1, 1, 2.0, -1.0), (512, 512, 1812, 0, 1, 1.0, 0.0), (448, 512, 1050, 1, 0, 1.0, 1.0)]:
    A, B, O0 = ints(M, K), ints(K, N), rng.integers(-3, 4, (M, N)).astype(np.float32)
    want = al * (A.astype(np.float64) @ B.astype(np.float64)) + be * O0        # float64 BLAS: exact on these integers, and fast
    dA, dB, dO = up(A.T if tA else A), up(B.T if tB else B), up(O0)
    busy()
    k.call("t4k_gemm", p(dA), p(dB), p(dO), al, be, tA, tB, M, N, K, 1, None)
    rc = k.lib.t4k_sync(None)
    if rc != 0 or not np.array_equal(dO.cpu().numpy().astype(np.float64), want.astype(np.float64)):
        fails.append(("gemm", M, N, K, tA, tB, rc, k.lib.t4k_last_error()))
for (Nb, E0, E1) in [(128, 100, 980), (128, 10, 100), (256, 512, 784), (256, 1, 256), (256, 256, 512), (64, 300, 1000)]:
    X, W, b, dY = ints(Nb, E1), ints(E0, E1), ints(E0), ints(Nb, E0)
    X64, W64, dY64 = X.astype(np.float64), W.astype(np.float64), dY.astype(np.float64)
    Y = (X64 @ W64.T + b).astype(np.int64)
    dW = (dY64.T @ X64).astype(np.int64); dBv = dY.sum(0).astype(np.int64); dX = (dY64 @ W64).astype(np.int64)
    dXd, dWd, dbd, dYd, dYo = up(X), up(W), up(b), up(dY), torch.zeros(Nb, E0, device="cuda")
    busy()
    k.call("t4k_linear_fwd", p(dXd), p(dWd), p(dbd), p(dYo), Nb, E0, E1, None)
    gW, gB = torch.ones(E0, E1, device="cuda"), torch.ones(E0, device="cuda")            # gradients ACCUMULATE
    k.call("t4k_linear_bwd", p(dXd), p(dWd), p(dYd), p(dXd), p(gW), p(gB), Nb, E0, E1, 1, None)       # dX over X: the in-place / gated path
    rc = k.lib.t4k_sync(None)
    ok = (rc == 0 and np.array_equal(dYo.cpu().numpy().astype(np.int64), Y) and np.array_equal(gW.cpu().numpy().astype(np.int64), dW + 1)
          and np.array_equal(gB.cpu().numpy().astype(np.int64), dBv + 1) and np.array_equal(dXd.cpu().numpy().astype(np.int64), dX))
    if not ok:
        fails.append(("linear", Nb, E0, E1, rc, k.lib.t4k_last_error()))
# the one-launch head backward (k_head_bwd_l32: dW1 || dX1 tiles that recompute dY1, column riders, the target-store rider - three kinds of workgroups
# that wait for each other through epoch-tagged slots) on integer operands, twice in a row
for (Nb, E1, EA, EB) in [(128, 980, 100, 10), (64, 512, 64, 16), (160, 256, 64, 10)]:
    if k.lib.t4k_mlp_head_bwd_ok(Nb, E1, EA, EB) != 1:
        continue                                              # (gates off: the host takes the two-launch path)
    X1, W1, X2, W2 = ints(Nb, E1), ints(EA, E1), ints(Nb, EA), ints(EB, EA)
    Pm, Tm = rng.integers(0, 3, (Nb, EB)).astype(np.float32), rng.integers(0, 2, (Nb, EB)).astype(np.float32)
    Mk = rng.integers(0, 2, (Nb, EA)).astype(np.float32)
    G2 = (Pm - Tm).astype(np.float64); DX2 = G2 @ W2.astype(np.float64); Y1 = DX2 * Mk
    want = {"P": G2, "Y2": G2, "X2": DX2, "Y1": Y1, "DW2": 1 + G2.T @ X2.astype(np.float64), "DB2": 1 + G2.sum(0),
            "DW1": 1 + Y1.T @ X1.astype(np.float64), "DB1": 1 + Y1.sum(0), "X1": Y1 @ W1.astype(np.float64)}
    for rep in range(2):
        d = {n_: up(v) for n_, v in dict(X1=X1, W1=W1, X2=X2, W2=W2, P=Pm, T=Tm, M=Mk).items()}
        for n_, shp in (("Y1", (Nb, EA)), ("Y2", (Nb, EB))): d[n_] = torch.zeros(shp, device="cuda")
        for n_, shp in (("DW1", (EA, E1)), ("DB1", (EA,)), ("DW2", (EB, EA)), ("DB2", (EB,))): d[n_] = torch.ones(shp, device="cuda")     # gradients ACCUMULATE
        busy()
        rc = k.lib.t4k_mlp_head_bwd(p(d["X2"]), p(d["W2"]), p(d["P"]), p(d["T"]), p(d["Y2"]), p(d["M"]), p(d["Y1"]), p(d["DW2"]), p(d["DB2"]),
                                    p(d["X1"]), p(d["W1"]), p(d["DW1"]), p(d["DB1"]), Nb, E1, EA, EB, None)
        rc = rc or k.lib.t4k_sync(None)
        bad = [n_ for n_, w_ in want.items() if not np.array_equal(d[n_].cpu().numpy().astype(np.float64), w_)]
        if rc != 0 or bad:
            fails.append(("head_bwd", Nb, E1, EA, EB, rep, rc, bad, k.lib.t4k_last_error()))
torch.cuda.synchronize()
print("FAILS", fails)
sys.exit(1 if fails else 0)
'''

# Round 6: the engine-selection knobs are LAB switches - compile-time constants in the release library (csrc/t4k_common.h T4K_LAB_ENV), read from the
# environment only by `make -C tensorforth_amd/csrc LAB=1` (libt4hip_lab.so).  The release suite runs the products on the DEFAULT dispatch (+ the busy-device
# case); the switch matrix runs against the LAB library when that has been built and T4K_TEST_LAB=1 asks for it (a lab bench, not part of the driver's suite).
LAB_LIB = os.path.join(ROOT, "tensorforth_amd", "libt4hip_lab.so")
LAB_SWITCHES = [{"T4K_LINTHIN": "0"}, {"T4K_LINTHIN_CW": "0"}, {"T4K_LINTHIN_CW": "16"}, {"T4K_GEMM_XMAP": "1"},
                {"T4K_GEMM_DUAL": "0"}, {"T4K_GEMM_DUAL32": "0"}, {"T4K_GEMM_S32": "0"}, {"T4K_GEMM_DUAL": "0", "T4K_GEMM_DUAL32": "0", "T4K_GEMM_S32": "0"},
                {"T4K_GEMM_FULLK": "0"}, {"T4K_GEMM_FASTPRO": "0"}, {"T4K_GEMM_RAGGED_DMA": "0"}, {"T4K_GEMM_PLAIN_BIG": "0"}, {"T4K_GEMM_BIG_DMA": "0"},
                {"T4K_GEMM_BIG_FULLK": "0"}, {"T4K_GEMM_DUAL_FULL": "0"}, {"T4K_GEMM_DUAL_FULLK": "0"}, {"T4K_GEMM_SPLIT_DIV": "2"},
                {"T4K_GEMM_S32_MAXK": "256"}, {"T4K_GEMM_DUAL_MAXK": "256"}, {"T4K_HEAD_FOLD": "0"}, {"T4K_LINSMALL_GATE": "0"}, {"T4K_LINSMALL_COLS": "0"}, {"T4K_LINSMALL_COLS": "2"},
                {"T4K_GEMM_PLAIN_PAIR": "0"}, {"T4K_GEMM_PLAIN128": "0"}, {"T4K_GEMM_PLAIN128": "2"}, {"T4K_GEMM_PLAIN128_RAGK": "0"}, {"T4K_GEMM_PLAIN128_BK32": "0"}, {"T4K_GEMM_PLAIN128_BK32": "2"},
                {"T4K_GEMM_PLAIN256": "0"}, {"T4K_GEMM_PLAIN256": "2"}, {"T4K_GEMM_PLAIN_RAGK": "0"}, {"T4K_GEMM_PLAIN_RAGK": "1"}, {"T4K_GEMM_PLAIN_ANY": "0"},
                {"T4K_GEMM_RAGGED_K": "0"}, {"T4K_GEMM_RAGGED_K": "2"}, {"T4K_GEMM_RAGGED_K": "2", "T4K_GEMM_S32": "0"}]
lab = pytest.mark.skipif(not (os.path.exists(LAB_LIB) and os.environ.get("T4K_TEST_LAB") == "1"), reason="LAB switch matrix: make -C tensorforth_amd/csrc LAB=1 and T4K_TEST_LAB=1")


def _run(tmp_path, env_extra):
    f = tmp_path / "sw.py"; f.write_text(_SCRIPT)
    env = dict(os.environ, T4_ROOT=ROOT); env.update(env_extra)
    r = subprocess.run([sys.executable, str(f)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, "%s\n%s\n%s" % (env_extra, r.stdout[-3000:], r.stderr[-2000:])


def test_integer_exact_products_on_the_default_dispatch(tmp_path):
    _run(tmp_path, {})


def test_gated_kernels_while_another_stream_hogs_the_device(tmp_path):
    _run(tmp_path, {"HOG": "1"})


def test_integer_exact_products_with_the_gated_kernels_switched_off(tmp_path):
    """t4k_gates_enable(0) (what the library does by itself after a timed-out wait): no kernel whose workgroups wait for each other is chosen."""
    _run(tmp_path, {"GATES_OFF": "1"})


@lab
@pytest.mark.parametrize("sw", LAB_SWITCHES, ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
def test_lab_integer_exact_products_under_every_dispatch_switch(tmp_path, sw):
    _run(tmp_path, dict(sw, T4K_LIB=LAB_LIB))


@lab
def test_lab_conv_parity_under_the_conv_engine_switches():
    """k_convbig8 / k_conv_thin_* variants (csrc/conv_big.hip, conv_img.hip): the conv parity tests under every LAB switch of the conv engines."""
    for env in ({"T4K_CONVBIG8": "0"}, {"T4K_CONVBIG8_BK32": "2"}, {"T4K_CONVBIG8_BK32": "0"}, {"T4K_CONVBIG8_NT": "1"}, {"T4K_CONVBIG_DFW": "0"}, {"T4K_CONVBIG_DFW": "3"}, {"T4K_CONVBIG_DFW": "4"},
                {"T4K_CONV_BN_RIDER": "0"}, {"T4K_BN_PART4": "0"}, {"T4K_CONV_THIN": "0"}, {"T4K_CONV_THIN_WG": "1"}, {"T4K_CONV_THIN_WG": "100000"}, {"T4K_CONV_THIN_NT": "0"},
                {"T4K_CONV_DF_WG": "2048", "T4K_CONV_THIN_DF": "0"}, {"T4K_CONV_THIN_DF_WG": "1"}, {"T4K_CONV_THIN_DF_WG": "100000"}):
        e = dict(os.environ, T4K_LIB=LAB_LIB, **env)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu",
                            "-k", "test_conv2d or many_channels or random_shapes or second_destination or batchnorm"], capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
        assert r.returncode == 0, str(env) + r.stdout[-3000:] + r.stderr[-2000:]
