"""Worker process around oracle/libten4_oracle.so - the CPU oracle VM behind the embedding API of include/ten4.h.

Test infrastructure only (the oracle is the checker, never the product).  It runs in a process of its own so that the oracle's
t4k_* / t4:: symbols can never meet the product's (libt4hip.so / libten4.so) in one address space.  Protocol on stdin/stdout:
length-prefixed pickles, one request -> one reply; requests are tuples ("eval", src) ("fetch", expr) ("tell",) ("seek", off)
("store", expr, ndarray) ("shard", rank, world) ("slab",) ("slab_set", ndarray) ("quit",)."""
import ctypes
import os
import pickle
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    seed = int(sys.argv[1])
    so = ctypes.CDLL(os.path.join(ROOT, "oracle", "libten4_oracle.so"))
    so.ten4_new.restype = ctypes.c_void_p
    so.ten4_new.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int]
    so.ten4_eval.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    so.ten4_output.restype = ctypes.c_char_p
    so.ten4_output.argtypes = [ctypes.c_void_p]
    so.ten4_rand_tell.restype = ctypes.c_ulonglong
    so.ten4_rand_tell.argtypes = [ctypes.c_void_p]
    so.ten4_rand_seek.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong]
    so.ten4_fetch.restype = ctypes.c_long
    so.ten4_fetch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.POINTER(ctypes.c_int * 4)]
    so.ten4_store.restype = ctypes.c_long
    so.ten4_store.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    so.ten4_grad_slab.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_long)]
    so.t4k_rand_set_shard.argtypes = [ctypes.c_int, ctypes.c_int]
    h = so.ten4_new(-1, seed, 0)
    assert h, "oracle VM failed to start"
    rd, wr = sys.stdin.buffer, sys.stdout.buffer

    def slab():
        p = ctypes.c_void_p(); n = ctypes.c_long()
        assert so.ten4_grad_slab(h, ctypes.byref(p), ctypes.byref(n)) == 0, "no finalized model"
        return np.ctypeslib.as_array(ctypes.cast(p.value, ctypes.POINTER(ctypes.c_float)), shape=(n.value,))

    def fetch():
        shp = (ctypes.c_int * 4)()
        n = so.ten4_fetch(h, None, 0, ctypes.byref(shp))
        if n < 0:
            return None
        a = np.empty(n, np.float32)
        so.ten4_fetch(h, a.ctypes.data_as(ctypes.c_void_p), n, ctypes.byref(shp))
        H, W, C, N = shp
        return a.reshape(N, H, W, C) if n == N * H * W * C else a

    while True:
        hdr = rd.read(8)
        if len(hdr) < 8:
            break
        req = pickle.loads(rd.read(struct.unpack("<Q", hdr)[0]))
        op = req[0]
        if op == "eval":
            so.ten4_eval(h, req[1].encode()); rep = so.ten4_output(h).decode(errors="replace")
        elif op == "fetch":
            txt = ""
            if req[1]:
                so.ten4_eval(h, req[1].encode()); txt = so.ten4_output(h).decode(errors="replace")
            rep = (fetch(), txt)
        elif op == "store":
            if req[1]:
                so.ten4_eval(h, req[1].encode()); so.ten4_output(h)
            a = np.ascontiguousarray(req[2], np.float32).ravel()
            rep = int(so.ten4_store(h, a.ctypes.data_as(ctypes.c_void_p), a.size))
        elif op == "tell":
            rep = int(so.ten4_rand_tell(h))
        elif op == "seek":
            so.ten4_rand_seek(h, int(req[1])); rep = None
        elif op == "shard":
            rep = so.t4k_rand_set_shard(int(req[1]), int(req[2]))
        elif op == "slab":
            rep = slab().copy()
        elif op == "slab_set":
            slab()[:] = req[1]; rep = None
        elif op == "quit":
            break
        else:
            rep = RuntimeError("bad request %r" % (op,))
        blob = pickle.dumps(rep)
        wr.write(struct.pack("<Q", len(blob))); wr.write(blob); wr.flush()


if __name__ == "__main__":
    main()
