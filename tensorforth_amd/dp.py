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
