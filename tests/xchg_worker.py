"""One RANK of a data-parallel job that shares ONE GPU with its peers (tests/test_gpu_xchg.py): a product VM training its rows of the
whole batch, gradients summed over the ranks by the one-shot peer exchange INSIDE the optimizer launch (csrc/xchg.hip).  The launcher side
channel is a directory: every rank writes its 64-byte window handle, waits for the others, connects.
usage: xchg_worker.py <dir> <rank> <world> <rows_per_rank> <steps> [absent]       (absent: this rank connects and leaves - the others must time out)"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    d, rank, world, rows, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    absent = len(sys.argv) > 6 and sys.argv[6] == "absent"
    from tensorforth_amd import lib as t4lib
    from tensorforth_amd.vm import VM
    from lenet_parity import PARAMS, _get, _setup
    k = t4lib.load()
    vm = VM(device=0, seed=505)
    _setup(vm, rows, rank * rows, world * rows)
    h = (ctypes.c_ubyte * 64)()
    k.call("t4k_xchg_create", 1 << 17, rank, world, h)
    with open(os.path.join(d, "h%d.tmp" % rank), "wb") as f:
        f.write(bytes(h))
    os.rename(os.path.join(d, "h%d.tmp" % rank), os.path.join(d, "h%d.bin" % rank))
    t0 = time.time()
    while not all(os.path.exists(os.path.join(d, "h%d.bin" % r)) for r in range(world)):
        assert time.time() - t0 < 120, "peers never showed up"
        time.sleep(0.01)
    allh = b"".join(open(os.path.join(d, "h%d.bin" % r), "rb").read() for r in range(world))
    k.call("t4k_xchg_connect", allh)
    assert k.lib.t4k_comm_world() == world and k.lib.t4k_comm_rank() == rank
    open(os.path.join(d, "c%d" % rank), "w").close()
    while not all(os.path.exists(os.path.join(d, "c%d" % r)) for r in range(world)):     # nobody pushes into a window that is not mapped yet
        time.sleep(0.01)
    if absent:
        time.sleep(20.0)                                    # stays alive (its window stays mapped) but never reaches the optimizer
        return 0
    k.lib.t4k_launch_count.restype = ctypes.c_ulonglong
    txt = vm.eval("net fw bw opt drop\n")                   # warm: the conv-stack kernels are loaded / compiled outside the count
    if "timed out" in txt or "failed" in txt:
        print("SYNC_ERROR in warm step: " + txt[-600:], flush=True)
        return 3
    rc = k.lib.t4k_sync(None)
    if rc != 0:
        print("SYNC_ERROR %d %s" % (rc, k.lib.t4k_last_error().decode(errors="replace")), flush=True)
        return 3
    l0 = k.lib.t4k_launch_count()
    for _ in range(steps - 1):
        txt = vm.eval("net fw bw opt drop\n")
        if "timed out" in txt or "failed" in txt:            # an inter-workgroup wait gave up inside the VM (it reports and goes on): this run is void
            print("SYNC_ERROR in step: " + txt[-600:], flush=True)
            return 3
    per = (k.lib.t4k_launch_count() - l0) / max(steps - 1, 1)
    rc = k.lib.t4k_sync(None)
    if rc != 0:
        print("SYNC_ERROR %d %s" % (rc, k.lib.t4k_last_error().decode(errors="replace")), flush=True)
        return 3
    out = {n_: _get(vm, e) for n_, e in PARAMS}
    txt = vm.eval("net fw lbl loss.ce . nn.hit . drop\n")  # whole-batch loss and hit count: scalar all-reduces over the same windows
    out["loss_hit"] = np.array([float(t) for t in txt.split()[:2]], np.float64)
    np.savez(os.path.join(d, "out%d.npz" % rank), launches=per, **out)
    k.call("t4k_xchg_destroy")
    return 0


if __name__ == "__main__":
    sys.exit(main())
