"""One-process-per-GPU launcher used by bench.py (and testable on CPU).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (the launch contract of the
task brief), after checking that the box really has N devices: a job that asked for 8 ranks must run 8 ranks or fail -
never print a 1-rank record labelled otherwise.  torch.distributed.run is plumbing (rendezvous + env), nothing else.
"""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """Devices this process could use (0 without a GPU / without the ROCm runtime)."""
    try:
        import torch
        return int(torch.cuda.device_count()) if torch.cuda.is_available() else 0
    except Exception:
        return 0


def need_spawn(n_gpus, environ=None):
    """True when the caller asked for several ranks but was started as a plain single process."""
    env = os.environ if environ is None else environ
    return n_gpus > 1 and "WORLD_SIZE" not in env and "RANK" not in env


def check_world(n_gpus, environ=None):
    """The launcher's world must be the one asked for; returns (rank, world, local_rank) or raises SystemExit(2)."""
    env = os.environ if environ is None else environ
    world = int(env.get("WORLD_SIZE", "1")); rank = int(env.get("RANK", "0")); local = int(env.get("LOCAL_RANK", str(rank)))
    if world != n_gpus:
        sys.stderr.write("bench: started with WORLD_SIZE=%d but --gpus %d: launch with `python bench.py --gpus %d` (self-spawning) or "
                         "`python -m torch.distributed.run --nproc-per-node %d ... bench.py --gpus %d`\n" % (world, n_gpus, n_gpus, n_gpus, n_gpus))
        raise SystemExit(2)
    return rank, world, local


def spawn_ranks(script, n_gpus, argv, dry_run=False, timeout=None):
    """Re-run `script argv` as n_gpus ranks on this node; returns the launcher's exit code (non-zero if any rank failed)."""
    if not dry_run:
        have = visible_gpus()
        if have < n_gpus:
            sys.stderr.write("bench: --gpus %d but only %d GPU(s) visible on this node: refusing to run fewer ranks than asked for\n" % (n_gpus, have))
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    return subprocess.call(cmd, env=env, timeout=timeout)
