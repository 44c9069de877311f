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
