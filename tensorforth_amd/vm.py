"""In-process binding of the host VM (libten4.so, include/ten4.h).

`VM.eval(src)` feeds Forth source to the same outer interpreter the stand-alone `ten4` REPL runs and
returns what it printed.  For data-parallel training `grad_slab()` exposes the model's contiguous
dW|dB buffer as a zero-copy torch view (CUDA array interface) so torch.distributed can all-reduce it
over RCCL, and `stream()` wraps the VM's HIP stream so the collective is ordered with the kernels.
torch is used for exactly that plumbing; all compute goes through libt4hip.so.
"""
import ctypes
import os

from . import lib as _lib

_HERE = os.path.dirname(os.path.abspath(__file__))


class _DevArray:
    """Minimal CUDA-array-interface holder around a raw device pointer (zero-copy import into torch)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2, "strides": None}


def set_lazy_dx0(on):
    """Process-wide launch-plan switch (ten4_set_lazy_dx0, the run-time form of T4_LAZY_DX0); returns the previous setting."""
    _lib.load()
    so = ctypes.CDLL(os.path.join(_HERE, "libten4.so"))
    so.ten4_set_lazy_dx0.restype = ctypes.c_int
    so.ten4_set_lazy_dx0.argtypes = [ctypes.c_int]
    return so.ten4_set_lazy_dx0(1 if on else 0)


class VM:
    def __init__(self, device=0, seed=1234, trace=0):
        _lib.load()                                      # imports torch first, then libt4hip.so (one HIP runtime)
        path = os.path.join(_HERE, "libten4.so")
        if not os.path.exists(path):
            raise RuntimeError("libten4.so not built: run __graft_entry__.build()")
        self._so = ctypes.CDLL(path)
        so = self._so
        so.ten4_new.restype = ctypes.c_void_p
        so.ten4_new.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int]
        so.ten4_free.argtypes = [ctypes.c_void_p]
        so.ten4_eval.restype = ctypes.c_int
        so.ten4_eval.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        so.ten4_output.restype = ctypes.c_char_p
        so.ten4_output.argtypes = [ctypes.c_void_p]
        so.ten4_grad_slab.restype = ctypes.c_int
        so.ten4_grad_slab.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_long)]
        so.ten4_stream.restype = ctypes.c_void_p
        so.ten4_stream.argtypes = [ctypes.c_void_p]
        so.ten4_rand_tell.restype = ctypes.c_ulonglong
        so.ten4_rand_tell.argtypes = [ctypes.c_void_p]
        so.ten4_rand_seek.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong]
        so.ten4_rand_reseed.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong]
        so.ten4_fetch.restype = ctypes.c_long
        so.ten4_fetch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.POINTER(ctypes.c_int * 4)]
        so.ten4_store.restype = ctypes.c_long
        so.ten4_store.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        self._HOOK = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_long, ctypes.c_long, ctypes.c_void_p)
        so.ten4_set_grad_hook.argtypes = [ctypes.c_void_p, self._HOOK, ctypes.c_void_p]
        self._hook_ref = None
        self._h = so.ten4_new(device, seed, trace)
        if not self._h:
            raise RuntimeError("ten4_new failed: no MI355X visible (the VM has no CPU fallback)")
        self.device = device

    def eval(self, src):
        """Run Forth source; returns the text the VM printed."""
        self._so.ten4_eval(self._h, src.encode())
        return self._so.ten4_output(self._h).decode(errors="replace")

    def rand_tell(self):
        """Position (in elements) of this VM's Philox stream."""
        return int(self._so.ten4_rand_tell(self._h))

    def rand_seek(self, off):
        self._so.ten4_rand_seek(self._h, int(off))

    def rand_reseed(self, seed):
        """give the VM a new Philox stream (seed, position 0): what a t4k_rand_init between evals cannot do (ten4.h)"""
        self._so.ten4_rand_reseed(self._h, int(seed))

    def fetch(self, expr=None):
        """Full-precision copy of the tensor `expr` leaves on top of the stack, as a numpy array shaped (N, H, W, C); the
        stack is left as `expr` left it (callers `drop` what they pushed)."""
        import numpy as np
        if expr:
            self.eval(expr)
        shp = (ctypes.c_int * 4)()
        n = self._so.ten4_fetch(self._h, None, 0, ctypes.byref(shp))
        if n < 0:
            raise RuntimeError("top of stack is not a tensor")
        a = np.empty(n, np.float32)
        self._so.ten4_fetch(self._h, a.ctypes.data_as(ctypes.c_void_p), n, ctypes.byref(shp))
        H, W, C, N = shp
        return a.reshape(N, H, W, C) if n == N * H * W * C else a

    def store(self, array, expr=None):
        """Fill the tensor `expr` leaves on top of the stack (a view from `nn.w`, `n@` ... writes through) with `array` (fp32)."""
        import numpy as np
        if expr:
            self.eval(expr)
        a = np.ascontiguousarray(array, np.float32).ravel()
        if self._so.ten4_store(self._h, a.ctypes.data_as(ctypes.c_void_p), a.size) != a.size:
            raise RuntimeError("top of stack is not a tensor of %d elements" % a.size)

    def grad_slab(self):
        """torch view (no copy) of the current model's gradient slab."""
        import torch
        p = ctypes.c_void_p()
        n = ctypes.c_long()
        if self._so.ten4_grad_slab(self._h, ctypes.byref(p), ctypes.byref(n)) != 0:
            raise RuntimeError("no finalized model: run `forward` once first")
        return torch.as_tensor(_DevArray(p.value, n.value), device="cuda:%d" % self.device)

    def stream(self):
        import torch
        return torch.cuda.ExternalStream(self._so.ten4_stream(self._h), device="cuda:%d" % self.device)

    def set_grad_hook(self, fn):
        """fn(layer, offset, n_floats) is called during `backprop` as each layer's slab segment becomes complete
        (stream-ordered); None removes the hook."""
        self._hook_ref = self._HOOK(lambda layer, off, n, _user: fn(layer, off, n)) if fn else self._HOOK(0)
        self._so.ten4_set_grad_hook(self._h, self._hook_ref, None)

    def close(self):
        if self._h:
            self._so.ten4_free(self._h)
            self._h = None
