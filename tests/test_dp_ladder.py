"""The fallback ladder of the gradient exchange (tensorforth_amd/dp.negotiate_reduction, what bench.py runs at N > 1) driven on the CPU: two ranks
over gloo and a STAND-IN for the library whose failures are scripted per rank (VERDICT r4 #9a - the hook is in the test, the product has no switch
that makes the exchange fail).  Whatever one rank sees, ALL ranks must land on the same transport, and the record must say which and why."""
import json
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class FakeLib:
    """the entry points negotiate_reduction touches; `fail` = {(entry, rank)} scripted failures, `corrupt` = rank whose exchange returns a wrong sum"""

    def __init__(self, rank, world, fail=(), corrupt=None):
        self.rank, self.world, self.fail, self.corrupt = rank, world, set(fail), corrupt
        self.calls = []
        self.connected = False

    def _rc(self, name):
        self.calls.append(name)
        return 1 if (name, self.rank) in self.fail or (name, "all") in self.fail else 0

    def t4k_comm_unique_id(self, raw):
        raw[0] = 42
        return self._rc("comm_unique_id")

    def t4k_comm_init(self, raw, rank, world):
        assert raw[0] == 42 and rank == self.rank and world == self.world
        return self._rc("comm_init")

    def t4k_comm_destroy(self): self.calls.append("comm_destroy"); return 0
    def t4k_comm_world(self): return self.world
    def t4k_comm_rank(self): return self.rank

    def t4k_xchg_create(self, n, rank, world, h):
        h[0] = 100 + rank
        return self._rc("xchg_create")

    def t4k_xchg_connect(self, blob):
        assert [blob[64 * r] for r in range(self.world)] == [100 + r for r in range(self.world)]      # every rank's handle, in rank order
        rc = self._rc("xchg_connect"); self.connected = rc == 0
        return rc

    def t4k_xchg_destroy(self): self.calls.append("xchg_destroy"); self.connected = False; return 0
    def t4k_xchg_world(self): return self.world if self.connected else 0

    def exchange(self, v):                                  # what t4k_xchg_allreduce does to the probe: an in-place SUM over the ranks
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        if self.corrupt == self.rank:
            v[12345] += 1.0
        return 0


def _worker(rank, world, port, out_dir, scenario):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tensorforth_amd import dp
    lib = FakeLib(rank, world, fail=[tuple(f) for f in scenario.get("fail", [])], corrupt=scenario.get("corrupt"))
    said = []
    res = dp.negotiate_reduction(lib, rank, world, None, log=said.append, xchg_allreduce=lib.exchange,
                                 want_native=scenario.get("want_native", True), want_xchg=scenario.get("want_xchg", True))
    with open(os.path.join(out_dir, "r%d.json" % rank), "w") as f:
        json.dump({"res": res, "calls": lib.calls, "said": said}, f)
    dist.destroy_process_group()


def _run(tmp_path, scenario, port):
    mp.spawn(_worker, args=(2, port, str(tmp_path), scenario), nprocs=2, join=True)
    return [json.load(open(os.path.join(str(tmp_path), "r%d.json" % r))) for r in range(2)]


def test_all_rungs_hold(tmp_path):
    r0, r1 = _run(tmp_path, {}, 29611)
    for r in (r0, r1):
        assert r["res"]["native"] and r["res"]["xchg"] and r["res"]["reason"] is None
        assert r["res"]["ranks_seen"] == {"torch.distributed": 2, "rccl": 2, "xchg": 2}


@pytest.mark.parametrize("scenario,why", [
    ({"fail": [["xchg_create", 1]]}, "receive window"),
    ({"fail": [["xchg_connect", 0]]}, "map a peer"),
    ({"corrupt": 1}, "self-check"),
], ids=["window-allocation-fails-on-rank-1", "ipc-mapping-fails-on-rank-0", "wrong-sum-on-rank-1"])
def test_exchange_failure_on_one_rank_sends_every_rank_to_rccl(tmp_path, scenario, why):
    r0, r1 = _run(tmp_path, scenario, 29612)
    for r in (r0, r1):
        assert r["res"]["native"] and not r["res"]["xchg"], r
        assert why in r["res"]["reason"] and "RCCL" in r["res"]["reason"]
        assert r["calls"][-1] == "xchg_destroy"                       # no rank keeps a half-connected exchange
        assert "xchg" not in r["res"]["ranks_seen"] and r["res"]["ranks_seen"]["rccl"] == 2


def test_communicator_failure_on_one_rank_sends_every_rank_to_torch_distributed(tmp_path):
    r0, r1 = _run(tmp_path, {"fail": [["comm_init", 1]]}, 29613)
    for r in (r0, r1):
        assert not r["res"]["native"] and not r["res"]["xchg"] and "torch.distributed" in r["res"]["reason"]
        assert "comm_destroy" in r["calls"] and not any(c.startswith("xchg") for c in r["calls"])     # the exchange is not even tried without the communicator


def test_switches_skip_rungs(tmp_path):
    r0, r1 = _run(tmp_path, {"want_xchg": False}, 29614)
    assert r0["res"]["native"] and not r0["res"]["xchg"] and not any(c.startswith("xchg") for c in r0["calls"] + r1["calls"])
