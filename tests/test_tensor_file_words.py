"""`save` / `load` of tensors (tenvm.cpp:597-598, AIO::tsave aio_tensor.cpp:75-93, _tsave_raw :240-255, _tsave_txt :230-238).

The raw layout is checked against bytes built HERE from the format's definition ('T','4', shape {H,W,C,N} as four U32, then one
byte per element = (U8)(v * 256)), not against another run of the host sources; the text form is the printer's with the 1024-cell
threshold.  CPU: the oracle-backed VM; GPU: the product VM (device tensors through t4k_memcpy)."""
import os
import struct

import numpy as np
import pytest

from vm_util import TEN4, TEN4_ORACLE, run_vm

VALS = [0.5, 0.25, 0.125, 0.75, 0.999, 0.0, 0.00390625, 0.0039, 0.3333333, 0.9999999, 0.1, 0.7]


def _script(d):
    lit = " ".join(repr(v) for v in VALS)
    return ("0 trace\n12 vector{ %s } 1 2 3 2 reshape4 constant t\n" % lit +
            # `save` / `load` ask for host service: like the reference (HOLD, eforth.h:85-92) the VM drops the rest of an interpreted line after them
            't s" %s/raw.t4" bin save\ndrop\n' % d +
            't s" %s/txt.txt" save\ndrop\n' % d +
            '1 2 3 2 tensor zeros s" %s/raw.t4" load\n." back " . \n' % d +
            '12 vector zeros s" %s/raw.t4" load\n." flat " . \n' % d +             # same element count, other shape: filled
            '5 vector ones s" %s/raw.t4" load\n." five " . \n' % d +               # element count differs: refused, tensor untouched
            '30 30 matrix ones 0.5 *= s" %s/big.txt" save\ndrop\n' % d +
            "bye\n")


def _check(out, d):
    raw = open(os.path.join(d, "raw.t4"), "rb").read()
    want = b"T4" + struct.pack("<4I", 2, 3, 2, 1) + bytes(int(np.float32(v) * 256.0) for v in VALS)
    assert raw == want
    txt = open(os.path.join(d, "txt.txt")).read()
    assert txt.split() == ("tensor[1,2,3,2] = { { { +0.5000_+0.2500 +0.1250_+0.7500 +0.9990_+0.0000 } "
                           "{ +0.0039_+0.0039 +0.3333_+1.0000 +0.1000_+0.7000 } } }").split()
    back = [int(np.float32(v) * 256.0) / 256.0 for v in VALS]
    toks = out.replace("_", " ").split()
    i = toks.index("back")
    got = [float(t) for t in toks[i:i + 40] if t[0] in "+-" and t[1].isdigit()][:12]
    np.testing.assert_allclose(got, back, atol=5.1e-5)                               # 4 printed decimals
    j = toks.index("flat")
    got = [float(t) for t in toks[j:j + 40] if t[0] in "+-" and t[1].isdigit()][:6]
    np.testing.assert_allclose(got, back[:3] + back[-3:], atol=5.1e-5)               # a 12-vector prints first / last three
    assert "element count differs" in out
    k = toks.index("five")
    assert [float(t) for t in toks[k:k + 12] if t[0] in "+-" and t[1].isdigit()][:5] == [1.0] * 5
    big = open(os.path.join(d, "big.txt")).read()
    assert "..." not in big and big.count("+0.5000") == 900                         # saved text is never elided below 1024 cells per side


@pytest.mark.skipif(not os.path.exists(TEN4_ORACLE), reason="oracle VM not built")
def test_tensor_save_load_words_on_the_oracle_vm(tmp_path):
    _check(run_vm(TEN4_ORACLE, source=_script(str(tmp_path))), str(tmp_path))


@pytest.mark.gpu
def test_tensor_save_load_words_on_the_product_vm(tmp_path):
    _check(run_vm(TEN4, source=_script(str(tmp_path))), str(tmp_path))
