"""Helpers for the script-level (Forth VM) parity tests.

`ten4` is the product (host VM over libt4hip.so, needs a GPU); `ten4_oracle` is the same host
sources linked against oracle/t4k_on_oracle.cpp (CPU, test infrastructure).  Both read Forth source
on stdin; outputs are compared token by token - words must match exactly, numbers within the
tolerance of the 4-decimal / 6-significant-digit formats the printer uses.
"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEN4 = os.path.join(ROOT, "tensorforth_amd", "ten4")
TEN4_ORACLE = os.path.join(ROOT, "oracle", "ten4_oracle")
SCRIPTS = os.path.join(ROOT, "tests", "scripts")
GOLDEN = os.path.join(ROOT, "tests", "golden", "vm")

_NUM = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)(e[+-]?\d+)?$|^[+-]?nan$|^[+-]?inf$", re.I)


def run_vm(binary, script_path=None, source=None, seed=1, timeout=300, env_extra=None, cwd=None):
    if source is None:
        with open(script_path) as f:
            source = f.read()
    env = dict(os.environ, T4_SEED=str(seed))
    if env_extra:
        env.update(env_extra)
    r = subprocess.run([binary], input=source, capture_output=True, text=True, timeout=timeout, env=env, cwd=cwd or ROOT)
    assert r.returncode == 0, f"{binary} rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}"
    return r.stdout


class OracleVM:
    """The CPU oracle VM with the interface of tensorforth_amd.vm.VM (eval / fetch / rand_tell / rand_seek), served by
    tests/oracle_vm_worker.py in its own process.  `fetch` returns FULL fp32 tensors, so parity is checked on every element
    at the north_star tolerance instead of on the printer's 4-decimal text."""

    def __init__(self, seed=1234):
        import sys
        if not os.path.exists(os.path.join(ROOT, "oracle", "libten4_oracle.so")):
            raise FileNotFoundError("oracle/libten4_oracle.so not built (make -C oracle)")
        self._p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "oracle_vm_worker.py"), str(seed)],
                                   stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=dict(os.environ, OMP_NUM_THREADS="1"))

    def _rpc(self, *req):
        import pickle
        import struct
        blob = pickle.dumps(req)
        self._p.stdin.write(struct.pack("<Q", len(blob))); self._p.stdin.write(blob); self._p.stdin.flush()
        hdr = self._p.stdout.read(8)
        assert len(hdr) == 8, "oracle VM worker died"
        rep = pickle.loads(self._p.stdout.read(struct.unpack("<Q", hdr)[0]))
        if isinstance(rep, Exception):
            raise rep
        return rep

    def eval(self, src):
        return self._rpc("eval", src)

    def fetch(self, expr=None):
        a, _txt = self._rpc("fetch", expr)
        if a is None:
            raise RuntimeError("top of stack is not a tensor")
        return a

    def store(self, array, expr=None):
        import numpy as np
        a = np.ascontiguousarray(array, np.float32)
        assert self._rpc("store", expr, a) == a.size, "top of stack is not a tensor of %d elements" % a.size

    def rand_tell(self):
        return self._rpc("tell")

    def rand_seek(self, off):
        self._rpc("seek", off)

    def set_shard(self, rank, world):
        assert self._rpc("shard", rank, world) == 0

    def grad_slab(self):
        return self._rpc("slab")

    def set_grad_slab(self, a):
        self._rpc("slab_set", a)

    def close(self):
        if self._p and self._p.poll() is None:
            try:
                self._rpc("quit")
            except Exception:
                pass
            self._p.stdin.close(); self._p.wait(timeout=10)
        self._p = None


def rel_err(got, want):
    """max |got - want| / max |want| over one tensor: the north_star "1e-4 relative" bar, per tensor and on every element."""
    import numpy as np
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    return float(np.max(np.abs(got - want)) / max(1e-30, np.max(np.abs(want))))


def elem_frac(got, want, rtol=1e-4, floor=1e-3):
    """ELEMENT-aware companion of rel_err (which is a tensor-norm figure: small elements get no bar of their own there): the share of
    elements with |got - want| <= rtol * (|want| + floor * max|want|) - every element is held to 1e-4 of ITS OWN magnitude, down to a
    floor of one thousandth of the tensor's largest element (below that an fp32 sum of ~1000 terms has no relative accuracy left, in the
    reference as much as here)."""
    import numpy as np
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    if want.size == 0:
        return 1.0
    bound = rtol * (np.abs(want) + floor * max(1e-30, np.max(np.abs(want))))
    return float(np.mean(np.abs(got - want) <= bound))


ELEM_MIN = 0.9999                                        # share of elements that must meet the element-aware bar (VERDICT r3 #5a)
ELEM_LOG = {}                                            # name -> worst share seen (printed by the tests that use it)


def check_tensor(name, got, want, tol=1e-4, elem_min=ELEM_MIN, floor=1e-3, min_outliers=0):
    """tensor-norm bar AND element-aware bar on one tensor.  floor = 1e-2 is used where the REFERENCE side is an fp32 sum of ~1e5 terms in
    sequential order (the oracle's conv filter gradients, nmath.tcu:211-338): measured against float64 the product meets the 1e-3 floor on
    every element of every LeNet gradient, the sequential sum does not (1 element of 90 / of 1800, tools/experiments/elemlog.py)."""
    e = rel_err(got, want)
    assert e <= tol, "%s: max|d|/max|ref| = %.3g > %.1g" % (name, e, tol)
    f = elem_frac(got, want, rtol=tol, floor=floor)
    ELEM_LOG[name] = min(f, ELEM_LOG.get(name, 1.0))
    import numpy as np
    n = int(np.asarray(want).size)
    allowed = max(min_outliers, int((1.0 - elem_min) * n + 1e-9))            # elements that may miss the element-aware bar (all of them still meet the tensor-norm bar above)
    assert round((1.0 - f) * n) <= allowed, "%s: %d of %d elements beyond %.0e of their own magnitude (floor %.0e max|ref|), %d allowed" % (name, round((1.0 - f) * n), n, tol, floor, allowed)


def tokens(text):
    out = []
    for line in text.splitlines():
        if line.startswith("tensorForth v4.0") or line.startswith("\\ MMU") or line.startswith("\\ "):
            continue                                    # banner / mstat lines name the backend
        for t in line.split():
            parts = t.split("_")                        # conv filters print as +a_+b_+c groups
            out.extend(parts if len(parts) > 1 and all(_NUM.match(q) for q in parts) else [t])
    return out


def compare(a_text, b_text, rtol=2e-4, atol=2.5e-4):
    """Return a list of mismatch descriptions (empty = parity)."""
    a, b = tokens(a_text), tokens(b_text)
    bad = []
    if len(a) != len(b):
        bad.append(f"token count {len(a)} != {len(b)}")
    for i, (x, y) in enumerate(zip(a, b)):
        if x == y:
            continue
        if _NUM.match(x) and _NUM.match(y):
            fx, fy = float(x), float(y)
            if fx != fx and fy != fy:
                continue
            if abs(fx - fy) <= atol + rtol * max(abs(fx), abs(fy)):
                continue
        bad.append(f"token {i}: {x!r} != {y!r} (context: {' '.join(a[max(0, i - 4):i + 3])})")
        if len(bad) > 10:
            break
    return bad


def numbers_after(text, label, count):
    """The `count` numbers that follow the first occurrence of `label` in the VM output."""
    toks = tokens(text)
    i = toks.index(label)
    vals = []
    for t in toks[i + 1:]:
        if _NUM.match(t):
            vals.append(float(t))
            if len(vals) == count:
                break
    return vals


def synth_mnist_dir(tmp_path_factory):
    """Working directory holding ./data/MNIST/raw with the synthetic MNIST-shaped corpus (seed 42)."""
    d = tmp_path_factory.mktemp("t4data")
    subprocess.run(["python3", os.path.join(ROOT, "tools", "make_synth_mnist.py"), os.path.join(str(d), "data", "MNIST", "raw"), "1024", "256"],
                   check=True, capture_output=True)
    # ... and ./data/CIFAR10/cifar-10-batches-bin with the synthetic CIFAR-10-shaped batches (seed 7 / 8)
    subprocess.run(["python3", os.path.join(ROOT, "tools", "make_synth_cifar.py"), os.path.join(str(d), "data", "CIFAR10", "cifar-10-batches-bin"), "256", "64"],
                   check=True, capture_output=True)
    # model files the REFERENCE's saver wrote (tests/golden/refhost/, tools/regen_vm_goldens.py): tests/scripts/model_load_ref.4th loads one by relative name
    import glob
    import shutil
    for f in glob.glob(os.path.join(ROOT, "tests", "golden", "refhost", "*.t4")):
        shutil.copy(f, str(d))
    return str(d)
