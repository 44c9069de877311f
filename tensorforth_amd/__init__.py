"""tensorforth_amd - MI355X (gfx950) tensor/CNN kernel backend behind tensorForth's words.

The product is ``libt4hip.so`` (hand-written HIP kernels behind the C-ABI of
``include/t4k.h``) plus the host VM ``ten4`` (C++).  This package only loads the library
for tests and ``bench.py``; there is no CPU fallback - loading fails loudly when the
library is missing and ``init()`` fails loudly when no gfx950 device is visible.
"""
from .lib import T4K, load, root_dir, T4KError  # noqa: F401
