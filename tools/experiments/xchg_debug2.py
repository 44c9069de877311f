"""which part of a data-parallel step goes wrong?  per step: local gradients of both ranks, parameters before / after the exchanging optimizer.
usage: xchg_debug2.py            (driver)      |  xchg_debug2.py worker <dir> <rank> <world>"""
import ctypes, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from lenet_parity import GRADS, PARAMS, _get, _setup
ROWS, STEPS = 32, 4
def worker(d, rank, world):
    from tensorforth_amd import lib as t4lib
    from tensorforth_amd.vm import VM
    k = t4lib.load(); vm = VM(device=0, seed=505)
    _setup(vm, ROWS, rank * ROWS, world * ROWS)
    h = (ctypes.c_ubyte * 64)(); k.call("t4k_xchg_create", 1 << 17, rank, world, h)
    open(os.path.join(d, "h%d.tmp" % rank), "wb").write(bytes(h)); os.rename(os.path.join(d, "h%d.tmp" % rank), os.path.join(d, "h%d.bin" % rank))
    while not all(os.path.exists(os.path.join(d, "h%d.bin" % r)) for r in range(world)): time.sleep(0.01)
    k.call("t4k_xchg_connect", b"".join(open(os.path.join(d, "h%d.bin" % r), "rb").read() for r in range(world)))
    open(os.path.join(d, "c%d" % rank), "w").close()
    while not all(os.path.exists(os.path.join(d, "c%d" % r)) for r in range(world)): time.sleep(0.01)
    out = {}
    if os.environ.get("DBG_FETCH", "1") == "1":
        for s in range(STEPS):
            vm.eval("net fw bw drop\n")
            for n_, e in GRADS: out["g%d_%s" % (s, n_)] = _get(vm, e)
            for n_, e in PARAMS: out["p%d_%s" % (s, n_)] = _get(vm, e)
            vm.eval("net opt drop\n")
            for n_, e in PARAMS: out["q%d_%s" % (s, n_)] = _get(vm, e)
    else:                                                    # no host sync inside the loop: device-side copies into stash tensors, fetched at the end
        src = ""
        for s in range(STEPS):
            src += "net fw bw drop\n"
            for n_, e in GRADS: src += "net %s copy constant g%d_%s drop drop\n" % (e, s, n_)
            for n_, e in PARAMS: src += "net %s copy constant p%d_%s drop drop\n" % (e, s, n_)
            src += "net opt drop\n"
            for n_, e in PARAMS: src += "net %s copy constant q%d_%s drop drop\n" % (e, s, n_)
        txt = vm.eval(src)
        assert "?" not in txt.replace("-> ok", ""), txt[-500:]
        for s in range(STEPS):
            for pre, lst in (("g", GRADS), ("p", PARAMS), ("q", PARAMS)):
                for n_, _e in lst:
                    out["%s%d_%s" % (pre, s, n_)] = vm.fetch("%s%d_%s" % (pre, s, n_)); vm.eval("drop")
    np.savez(os.path.join(d, "dbg%d.npz" % rank), **out)
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4])); sys.exit(0)
world = 2
for trial in range(6):
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", d, str(r), str(world)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        outs = [p.communicate(timeout=200)[0] for p in ps]
        if any(p.returncode for p in ps): print("FAILED", outs); continue
        R = [np.load(os.path.join(d, "dbg%d.npz" % r)) for r in range(world)]
        line = []
        for s in range(STEPS):
            for (gn, _), (pn, _) in zip(GRADS, PARAMS):
                if "g%d_%s" % (s, gn) not in R[0]: continue
                gsum = sum(R[r]["g%d_%s" % (s, gn)].astype(np.float32) for r in range(world))
                p0 = R[0]["p%d_%s" % (s, pn)]; q0 = R[0]["q%d_%s" % (s, pn)]
                nw = p0.shape[0]
                want = p0 - np.float32(0.01) * (gsum / np.float32(nw))
                err = np.abs(q0 - want).max() / max(1e-30, np.abs(want).max())
                same = np.array_equal(R[0]["q%d_%s" % (s, pn)], R[1]["q%d_%s" % (s, pn)])
                if err > 1e-5 or not same:
                    bad = np.argwhere(np.abs(q0 - want) > 1e-5 * np.abs(want).max())
                    line.append("s%d %s err=%.1e same=%s nbad=%d of %d first=%s" % (s, pn, err, same, len(bad), q0.size, bad[:3].tolist()))
        print("trial", trial, "OK" if not line else "; ".join(line[:6]), flush=True)
