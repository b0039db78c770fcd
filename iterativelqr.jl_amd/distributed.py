"""Batch sharding across the GPUs of one node, one process per GPU.

The path is embarrassingly parallel (no cross-trajectory coupling in the reference), so rank r owns the contiguous
instance range [r*B, (r+1)*B) and NO data-path collective exists; torch.distributed (backend "nccl" = RCCL over xGMI,
"gloo" in the CPU tests) carries only the timing barrier, a MAX over ranks and an all-gather of per-rank figures.

`launch_ranks` is the one-node launcher behind `bench.py --gpus N`: it starts N fresh child processes BEFORE the
parent has touched the GPU (never an exec of a process that has initialised HIP), with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the way `python -m torch.distributed.run` would set them.
"""
import os
import socket
import subprocess
import sys


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(rank, per_rank_batch):
    """Instance index range owned by `rank` (weak scaling: fixed batch per GPU)."""
    return rank * per_rank_batch, (rank + 1) * per_rank_batch


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_devices():
    """Number of GPUs this process could use, WITHOUT initialising HIP (torch.cuda.device_count() does not, on this
    image; torch.cuda.is_available() would)."""
    import torch
    return int(torch.cuda.device_count())


def launch_ranks(script, argv, nproc, share_device=False, stub=False, timeout=None):
    """Start `nproc` ranks of `script argv...` and wait. Returns (exit_code, stdout of rank 0).

    Fails loudly (RuntimeError) if fewer than nproc devices are visible, unless the ranks are told to share device 0
    (ILQR_BENCH_SHARE_DEVICE=1, a test hook for one-GPU boxes) or to run the CPU stub."""
    if not (share_device or stub):
        have = visible_devices()
        if have < nproc:
            raise RuntimeError("--gpus %d requested but only %d GPU(s) are visible to this process; refusing to "
                               "report a %d-GPU figure from fewer devices" % (nproc, have, nproc))
    port = free_port()
    procs = []
    for r in range(nproc):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's pipe is drained by a thread while ALL children are polled: a rank that dies at start-up (no GPU, import error)
    # must end the job at once — the others would sit in init_process_group / a barrier until the collective timeout
    import threading
    import time
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    deadline = None if timeout is None else time.time() + timeout
    try:
        while True:
            codes = [p.poll() for p in procs]
            failed = [c for c in codes if c not in (None, 0)]
            if failed:
                rc = failed[0]
                break
            if all(c == 0 for c in codes):
                break
            if deadline is not None and time.time() > deadline:
                rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:           # exactly the children started here, by PID
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    reader.join(timeout=5)
    out0 = b"".join(c for c in chunks if c)
    return rc, out0.decode()


def max_over_ranks(value, dist=None, device=None):
    """MAX-reduce a python float over the process group (identity without one)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def torch_allreduce_sum(dist, device=None):
    """allreduce_sum callable for Solver.solve_shared_step_: SUM over the process group of a short float64 vector
    (RCCL when the group's backend is nccl and `device` is the rank's GPU, gloo on CPU tensors otherwise)."""
    import numpy as np
    import torch

    def ar(v):
        if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
            return v
        t = torch.tensor(np.asarray(v, dtype=np.float64), dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()
    return ar


def gather_over_ranks(values, dist=None, device=None):
    """All-gather a short list of floats: returns [world][len(values)] (one row without a process group)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [[float(v) for v in values]]
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    outs = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [[float(v) for v in o.cpu()] for o in outs]
