"""Sample-sharded data parallelism for the nn words: one process per GPU, identical
replicas, each rank trains on its own rows of the global batch, and ONE collective per step - an
all-reduce(SUM) of the contiguous gradient slab - runs between `backprop` and the optimizer word.

SUM, not mean: the reference's optimizers consume raw batch-sum gradients (`k_sgd` divides by the
parameter tensor's N, not by the batch; gradient.cu:63-126), so summing the shards reproduces the
single-GPU update on the concatenated batch exactly.

torch.distributed is the transport (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU
tests); nothing here touches the compute path.
"""
import torch
import torch.distributed as dist


def shard_rows(global_batch, rank, world):
    """Rows [lo, hi) of the global batch that `rank` trains on (contiguous, equal shards)."""
    if global_batch % world:
        raise ValueError("global batch %d is not divisible by %d ranks" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def rank_rng_offset(rank, stride_log2=36):
    """Philox stream offset for a rank's private draws (synthetic shard, dropout masks); offset 0..2^36 is the
    part of the stream every replica shares (weight init), so all ranks start from identical parameters."""
    return (rank + 1) << stride_log2


def allreduce_grad_slab(slab, group=None, stream=None):
    """In-place SUM of the gradient slab over all ranks.  `stream` (a torch.cuda stream wrapping the
    VM's HIP stream) orders the collective after backprop's kernels and before the optimizer's."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return slab
    if stream is not None:
        with torch.cuda.stream(stream):
            dist.all_reduce(slab, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.all_reduce(slab, op=dist.ReduceOp.SUM, group=group)
    return slab


def allreduce_scalars(values, group=None, device=None):
    """SUM a few per-rank scalars (loss * N_local, hit count) in one small collective."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.tolist()


class OverlappedSlabReducer:
    """Gradient all-reduce overlapped with backprop.

    Layers finish their dW|dB in reverse order, and the slab is laid out in layer order, so the TAIL of the slab is
    complete first - for CNNs that is the fully-connected part, i.e. almost all of the bytes (97 % for the LeNet
    net).  `on_layer` is installed as the VM's gradient hook: as soon as the completed tail covers `tail_frac` of
    the slab it is all-reduced on a side stream (ordered after the VM stream by an event) while the convolution
    layers are still back-propagating; `finish()` reduces the small remaining head and makes the VM stream wait for
    both, so the optimizer word that follows sees the summed gradients."""

    def __init__(self, slab, vm_stream, group=None, tail_frac=0.5):
        self.slab, self.vs, self.group, self.tail_frac = slab, vm_stream, group, tail_frac
        self.side = torch.cuda.Stream(device=slab.device) if slab.is_cuda else None
        self.cut = None                      # slab[cut:] already reducing

    def on_layer(self, layer, off, n):
        if self.cut is not None or (self.slab.numel() - off) < self.tail_frac * self.slab.numel():
            return
        self.cut = off
        if self.side is None:                # CPU tensors (gloo tests): no streams
            dist.all_reduce(self.slab[off:], op=dist.ReduceOp.SUM, group=self.group)
            return
        ev = torch.cuda.Event()
        ev.record(self.vs)                   # everything backprop has enqueued so far
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            dist.all_reduce(self.slab[off:], op=dist.ReduceOp.SUM, group=self.group)

    def finish(self):
        head = self.slab.numel() if self.cut is None else self.cut
        if head > 0:
            if self.side is None:
                dist.all_reduce(self.slab[:head], op=dist.ReduceOp.SUM, group=self.group)
            else:
                with torch.cuda.stream(self.vs):
                    dist.all_reduce(self.slab[:head], op=dist.ReduceOp.SUM, group=self.group)
        if self.side is not None and self.cut is not None:
            self.vs.wait_stream(self.side)
        self.cut = None


def negotiate_reduction(lib, rank, world, device, want_native=True, want_xchg=True, log=None, xchg_allreduce=None, slab_floats=1 << 17):
    """The fallback ladder of the gradient exchange, agreed on by ALL ranks (every decision is an all-reduce(MIN) of the ranks' verdicts, so
    no rank is left on a transport its peers abandoned):

      1. the library's own RCCL communicator (`t4k_comm_*`; torch.distributed only carries the 128-byte id) - else torch.distributed reduces the slab;
      2. on top of it the one-shot peer exchange (`t4k_xchg_*`, csrc/xchg.hip): every rank's receive window is shared through an IPC handle, then
         CHECKED against a known sum on both window parities before it is trusted - else the slab goes through RCCL (rung 1).

    `lib` is the ctypes library (or, in the CPU tests, a stand-in with the same entry points whose failures are scripted: the ladder itself is
    transport-agnostic and runs over gloo there).  Returns {"native", "xchg", "reason", "ranks_seen"}; the caller prints `reason` and
    `ranks_seen` so that a driver can verify that N ranks really agreed."""
    import ctypes
    say = log or (lambda m: None)
    res = {"native": False, "xchg": False, "reason": None, "ranks_seen": {}}

    def all_min(flag):
        t = torch.tensor([1.0 if flag else 0.0], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item()) > 0

    def count_ranks():
        t = torch.ones(1, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(round(float(t.item())))

    res["ranks_seen"]["torch.distributed"] = count_ranks()
    if want_native:
        idbuf = torch.zeros(128, dtype=torch.uint8, device=device)
        have_id = True
        if rank == 0:
            raw = (ctypes.c_ubyte * 128)()
            if lib.t4k_comm_unique_id(raw) == 0:
                idbuf.copy_(torch.tensor(list(raw), dtype=torch.uint8))
            else:
                have_id = False
        dist.broadcast(idbuf, 0)
        if all_min(have_id):
            raw = (ctypes.c_ubyte * 128)(*idbuf.cpu().tolist())
            res["native"] = all_min(lib.t4k_comm_init(raw, rank, world) == 0)
            if not res["native"]:
                lib.t4k_comm_destroy()
                res["reason"] = "a rank could not join the library's RCCL communicator: torch.distributed reduces the slab"
            elif lib.t4k_comm_world() != world or lib.t4k_comm_rank() != rank:
                raise SystemExit("RCCL communicator has %d ranks (this one is %d), %d asked for" % (lib.t4k_comm_world(), lib.t4k_comm_rank(), world))
            else:
                res["ranks_seen"]["rccl"] = lib.t4k_comm_world()
        else:
            res["reason"] = "rank 0 could not create an RCCL id (librccl missing?): torch.distributed reduces the slab"
    if res["native"] and want_xchg and world > 1:
        h = (ctypes.c_ubyte * 64)()
        created = lib.t4k_xchg_create(slab_floats, rank, world, h) == 0
        mine = torch.tensor(list(h), dtype=torch.uint8, device=device)
        allh = [torch.zeros(64, dtype=torch.uint8, device=device) for _ in range(world)]
        dist.all_gather(allh, mine)
        good = all_min(created)
        why = "a rank could not allocate / export its receive window"
        if good:
            blob = b"".join(bytes(t.cpu().tolist()) for t in allh)
            good = all_min(lib.t4k_xchg_connect(blob) == 0)
            why = "a rank could not map a peer's window (hipIpcOpenMemHandle; only SOME peers mappable counts as none: the exchange is all-or-nothing)"
            dist.barrier()
        if good:                                              # self-check: sum over ranks of (rank + 1) * i must be i * world (world + 1) / 2, on both window parities
            probe = torch.arange(70000, dtype=torch.float32, device=device) % 1000
            run = xchg_allreduce or (lambda v: lib.t4k_xchg_allreduce(v.data_ptr(), v.numel(), None) or lib.t4k_sync(None))
            import time
            on_gpu = device is not None and str(device).startswith("cuda")
            res["xchg_probe_us"] = []                          # this rank's wall time of each probe call (the first carries first-use costs)
            for _ in range(2):
                v = (probe * (rank + 1)).contiguous()
                if on_gpu:
                    torch.cuda.synchronize()
                dist.barrier()                                 # the probe's time is the exchange's, not the wait for a rank that is still uploading
                t0 = time.perf_counter()
                rc = run(v)
                if on_gpu:
                    torch.cuda.synchronize()
                res["xchg_probe_us"].append(round((time.perf_counter() - t0) * 1e6, 1))
                good = all_min(rc == 0 and bool(torch.equal(v, probe * (world * (world + 1) // 2))))
                why = "the exchange failed its known-sum self-check"
                if not good:
                    break
        res["xchg"] = good
        if good:
            if hasattr(lib, "t4k_xchg_trust"):
                lib.t4k_xchg_trust(1)                          # from here on a late peer gets the full patience (T4K_XCHG_TIMEOUT_MS)
            res["ranks_seen"]["xchg"] = lib.t4k_xchg_world()
        else:
            res["reason"] = "one-shot peer exchange not used (%s) - the slab goes through RCCL" % why
            say(res["reason"])
            lib.t4k_xchg_destroy()
    return res
