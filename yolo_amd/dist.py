"""Batch sharding over the GPUs of one node: one process per GPU, no per-layer collectives.

Images are independent, weights are replicated, every rank runs conv -> decode -> NMS on its contiguous
slice of the batch.  The only exchange is the final gather of the fixed-size detection blocks
(det[B_local, max_det, 6] fp32 + count[B_local] int32, packed into one buffer) with a single all_gather
over RCCL (backend "nccl" on ROCm) - or gloo on the CPU in tests.  The reference has no counterpart:
its evaluation is single-GPU (test.py:74-76).
"""
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* when launched by
    torch.distributed.run; returns (rank, world_size, local_rank). A plain single-process run returns (0, 1, 0)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:  # Y5_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get('Y5_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def pin_cpus(local_rank, local_world):
    """Give this rank its own contiguous share of the CPUs the process may run on (sched_setaffinity): eight Python
    launch threads then do not migrate across sockets or share cores.  Contiguous shares follow the usual numbering
    (sockets in order, GPUs 0..n/2-1 on the first socket).  Y5_PIN_CPUS=0 leaves the affinity alone.  Returns the
    CPU set taken, or None when nothing was changed."""
    if os.environ.get('Y5_PIN_CPUS', '1') == '0' or local_world <= 1 or not hasattr(os, 'sched_setaffinity'):
        return None
    cpus = sorted(os.sched_getaffinity(0))
    share = len(cpus) // local_world
    if share < 1:
        return None
    mine = set(cpus[local_rank * share:(local_rank + 1) * share])
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine


def tune_once(warm, group=None):
    """Kernel-variant choices made once for the job: rank 0 runs warm() (the forwards whose launch plans time every
    variant), sends its choices to the other ranks, and those run warm() replaying them without timing anything.  All
    ranks then launch the same kernels (the per-step time is the MAX over ranks, so one rank's unlucky choice is the
    job's) and start-up costs one tuning pass, not `world` concurrent ones.  Ranks whose library build or device differs
    from rank 0's ignore the choices and tune for themselves.  Returns the number of choices this rank took over."""
    from yolo_amd import engine
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        warm()
        return 0
    rank = dist.get_rank(group)
    payload = [None]
    if rank == 0:
        warm()
        payload[0] = engine.export_tuning()
    dist.broadcast_object_list(payload, src=0, group=group)
    taken = 0
    if rank != 0:
        taken = engine.import_tuning(payload[0])
        warm()
    return taken


def shard_bounds(total, rank, world):
    """Contiguous slice [lo, hi) of a batch of `total` images owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_detections(det, count):
    """(B, max_det, 6) fp32 + (B,) int32 -> one (B, max_det*6 + 1) fp32 block (the count travels as raw bits)."""
    B = det.shape[0]
    block = torch.empty(B, det.shape[1] * 6 + 1, dtype=torch.float32, device=det.device)
    block[:, :-1] = det.reshape(B, -1)
    block[:, -1] = count.to(torch.int32).view(torch.float32)
    return block


def unpack_detections(block, max_det):
    det = block[:, :-1].reshape(block.shape[0], max_det, 6)
    count = block[:, -1].contiguous().view(torch.int32)
    return det, count


def gather_detections(det, count, group=None):
    """All ranks contribute equally sized blocks; every rank receives (det_all, count_all) in global image order
    (rank-major, which equals image order for contiguous shards)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return det, count
    world = dist.get_world_size(group)
    block = pack_detections(det, count)
    if block.is_cuda and dist.get_backend(group) == 'gloo':  # rehearsal on a CPU collective: stage through the host
        host = block.cpu()
        out = torch.empty((world * host.shape[0], host.shape[1]), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host, group=group)
        return unpack_detections(out.to(block.device), det.shape[1])
    out = torch.empty((world * block.shape[0], block.shape[1]), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(out, block, group=group)
    return unpack_detections(out, det.shape[1])


def to_list(det, count):
    """Padded detections -> the reference's list[None | Tensor(n, 6)] (utils/utils.py:584,641)."""
    counts = count.tolist()
    return [det[b, :c].clone() if c > 0 else None for b, c in enumerate(counts)]


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(argv, world, env=None, timeout=None):
    """Start `world` child processes running `argv` (a full command line, e.g. [sys.executable, 'bench.py', ...]), one
    per rank, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the way torch.distributed.run sets
    them, and wait for them.  Meant to be called BEFORE the calling process has touched a GPU: the parent only
    supervises.  Rank 0 inherits stdout (its one JSON line is the job's output), the other ranks' stdout is dropped,
    stderr is inherited by all.  When there are fewer GPUs than ranks the children are told to rehearse on the gloo
    backend (Y5_DIST_BACKEND) and share devices.  Returns 0 when every rank exited 0, else the first non-zero exit
    code; on a failure or a timeout the remaining children are terminated (by their own PIDs)."""
    base = dict(os.environ if env is None else env)
    base.setdefault('MASTER_ADDR', '127.0.0.1')
    base['MASTER_PORT'] = str(free_port())
    base['WORLD_SIZE'] = str(world)
    if 'Y5_DIST_BACKEND' not in base and torch.cuda.device_count() < world:  # device_count() does not initialise the GPU
        base['Y5_DIST_BACKEND'] = 'gloo'
    procs = []
    for rank in range(world):
        e = dict(base, RANK=str(rank), LOCAL_RANK=str(rank))
        out = None if rank == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen(list(argv), env=e, stdout=out))
    deadline = None if timeout is None else time.time() + timeout
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
        if rc != 0 or (deadline is not None and time.time() > deadline):
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
            return rc or 124
        time.sleep(0.05)
    return rc
