"""Loader for the product C-ABI library libt4hip.so (see include/t4k.h)."""
import ctypes
import os

from . import _cabi

_HERE = os.path.dirname(os.path.abspath(__file__))


def root_dir():
    return os.path.dirname(_HERE)


class T4KError(RuntimeError):
    pass


class T4K:
    """Thin handle on libt4hip.so: attribute access returns the bound C functions;
    ``call(name, *args)`` raises T4KError with t4k_last_error() on a non-zero status."""

    def __init__(self, path=None):
        path = path or os.path.join(_HERE, "libt4hip.so")
        if not os.path.exists(path):
            raise T4KError("libt4hip.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        self.path = path
        try:                      # one HIP runtime per process: if torch is around, let it load its
            import torch          # bundled libamdhip64 first so libt4hip.so binds to the same one
            torch.cuda.is_available()
        except Exception:
            pass
        self.lib = ctypes.CDLL(path)
        self.decls = _cabi.parse_header(os.path.join(root_dir(), "include", "t4k.h"), "t4k_")
        self.missing = _cabi.bind(self.lib, self.decls)

    def __getattr__(self, name):
        return getattr(self.lib, name)

    def call(self, name, *args):
        rc = getattr(self.lib, name)(*args)
        if rc != 0:
            raise T4KError("%s failed (%d): %s" % (name, rc, self.lib.t4k_last_error().decode()))
        return rc

    def init(self, device=0):
        self.call("t4k_init", device)


_singleton = None


def load(path=None):
    global _singleton
    if _singleton is None or path:
        _singleton = T4K(path)
    return _singleton
