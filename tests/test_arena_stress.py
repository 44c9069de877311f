"""Allocator (host/tensor.cpp `Arena`: slabs from the backend, host-side free lists, best fit, coalescing; replaces src/mu/tlsf.cpp +
mmu.cu's object store).  Stress through the VM's own words on the oracle-backed VM (same Arena code, host memory instead of HBM):
10^5 allocate / free pairs of mixed sizes with long-lived tensors interleaved, a fragmentation pattern (a freed block between every two survivors, then
the same sizes again: the holes must be reused), and slab growth (T4_SLAB_MB=1 forces new slabs).  `mstat` reports bytes in use, live blocks, peak, free
blocks and slabs: after everything is dropped the arena must be back to zero bytes in ONE free block per slab (full coalescing), reuse
must not grow the slab count, and the whole run must finish in seconds (the size index makes allocation O(log n))."""
import os
import re
import subprocess
import time

import pytest

from vm_util import TEN4, TEN4_ORACLE

SCRIPT = r'''0 trace
: churn ( n -- ) 0 do i 13 mod 1 + 64 * vector drop loop ;
: mixed ( n -- ) 0 do i 5 mod 1 + 100 * i 3 mod 1 + 10 * matrix i 7 mod 1 + 32 * vector drop drop loop ;
mstat
100000 churn
." after_churn " cr
mstat
30000 mixed
." after_mixed " cr
mstat
\ long-lived tensors interleaved with garbage: 64 survivors of growing size
: hold ( -- t1 .. t64 ) 64 0 do i 1 + 256 * vector 1000 vector drop loop ;
hold
." holding " cr
mstat
\ fragmentation: survivors A with a freed block B between each pair (B is allocated between two A's, then dropped from under the top)
: frag ( -- A0 .. A32 ) 256 vector 32 0 do i 1 + 300 * vector i 2 + 256 * vector swap drop loop ;
frag
." fragged " cr
mstat
\ holes of 300 .. 9600 floats are now scattered between live blocks: best fit must reuse them for same-size requests (no new slab)
: refill ( -- B0 .. B31 ) 32 0 do i 1 + 300 * vector loop ;
refill
." refilled " cr
mstat
: dropall ( .. n -- ) 0 do drop loop ;
129 dropall
." dropped " cr
mstat
\ a tensor larger than a slab gets a slab of its own
600000 vector drop
." big " cr
mstat
bye
'''


def _stats(out):
    res = {}
    # (label on its own line, `mstat` on the next: the reference's mstat is a plain printf, which on ONE line would come out in front of the buffered label)
    for m in re.finditer(r"(\w+) \n[^\n]*-> ok\n\\ MMU\.stat .*?obj#used\[(\d+)\], HBM used=(\d+) KiB in (\d+) blocks \(peak (\d+) KiB, (\d+) free block\(s\), (\d+) slab\(s\)\)", out):
        res[m.group(1)] = dict(objs=int(m.group(2)), kib=int(m.group(3)), blocks=int(m.group(4)), peak=int(m.group(5)), free=int(m.group(6)), slabs=int(m.group(7)))
    return res


def _run(binary):
    t0 = time.time()
    r = subprocess.run([binary], input=SCRIPT, capture_output=True, text=True, timeout=300, env=dict(os.environ, T4_SEED="1", T4_SLAB_MB="1"))
    dt = time.time() - t0
    assert r.returncode == 0 and "?" not in r.stdout.replace("-> ok", ""), r.stdout[-2000:]
    s = _stats(r.stdout)
    assert set(s) >= {"after_churn", "after_mixed", "holding", "fragged", "refilled", "dropped", "big"}, r.stdout[-3000:]
    # garbage loops leave nothing behind, and reuse means the 10^5 allocations never needed a second 1 MiB slab
    assert s["after_churn"]["kib"] == 0 and s["after_churn"]["blocks"] == 0 and s["after_churn"]["slabs"] == 1 and s["after_churn"]["free"] == 1
    assert s["after_mixed"]["kib"] == 0 and s["after_mixed"]["free"] == s["after_mixed"]["slabs"]            # fully coalesced: one free block per slab
    # 64 survivors: sum (i * 256 floats * 4 B) = 2080 KiB live -> slab growth (1 MiB slabs)
    assert s["holding"]["blocks"] == 64 and s["holding"]["kib"] == sum((i + 1) * 256 * 4 for i in range(64)) // 1024 and s["holding"]["slabs"] >= 3
    assert s["fragged"]["blocks"] == 64 + 33
    assert s["refilled"]["blocks"] == 64 + 33 + 32 and s["refilled"]["slabs"] <= s["fragged"]["slabs"] + 1     # holes are reused (later survivors already took some)
    assert s["dropped"]["kib"] == 0 and s["dropped"]["blocks"] == 0 and s["dropped"]["free"] == s["dropped"]["slabs"]
    assert s["big"]["slabs"] == s["dropped"]["slabs"] + 1 and s["big"]["kib"] == 0                          # 2.4 MB tensor: a slab of its own (4 MiB), returned whole
    assert dt < 60, "allocator too slow: %.1f s" % dt
    return s


@pytest.mark.skipif(not os.path.exists(TEN4_ORACLE), reason="oracle VM not built")
def test_arena_stress_on_the_oracle_vm():
    _run(TEN4_ORACLE)


@pytest.mark.gpu
def test_arena_stress_on_hbm():
    _run(TEN4)
