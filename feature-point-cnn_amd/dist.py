"""Frame-batch sharding over one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The path has no cross-frame state (BatchNorm uses running statistics; NMS, sort
and descriptor sampling are per frame), so a batch of frames shards contiguously
across ranks with NO data-path collective.  The only exchange is at start-up:
rank 0 parses the checkpoint, folds and packs it, and broadcasts the packed blob
(a few MB) so the other ranks never touch the file.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).
    Returns (rank, world_size, local_rank).  A single process needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # FPC_DIST_BACKEND=gloo: rehearse the N>1 path with several ranks on ONE GPU (RCCL refuses
            # two ranks on the same device); the data path has no collective, so only start-up differs
            backend = os.environ.get("FPC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_frames, world, rank):
    """Contiguous shard [lo, hi) of frame indices for `rank`:
    frame f -> rank f // ceil(n / world), remainders to the low ranks."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_packed_weights(engine, state_dict, src=0):
    """Rank `src` loads `state_dict` into its engine; every other rank receives the packed,
    BN-folded blob.  With NCCL/RCCL the broadcast writes straight into the library's
    device buffer; with gloo it goes through host memory."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        engine.load_state_dict(state_dict)
        return
    rank = dist.get_rank()
    if rank == src:
        engine.load_state_dict(state_dict)
    if dist.get_backend() == "nccl":
        # RCCL broadcast over xGMI, device to device, of a torch-owned buffer (16 MB, start-up only), then one copy
        # into the library.  FPC_DIST_ZERO_COPY=1 broadcasts straight into the library's blob instead (a tensor view
        # of memory the library allocated): it saves that copy, but a failure on SOME ranks only would leave the ranks
        # in different collectives, so it is opt-in and every rank must set it.
        done = False
        if os.environ.get("FPC_DIST_ZERO_COPY") == "1":
            view = engine.packed_view()
            dist.broadcast(view, src=src)
            torch.cuda.synchronize()
            if rank != src:
                engine.mark_weights_loaded()
            done = True
        if not done:
            n = engine.packed_size()
            dev = torch.device("cuda", torch.cuda.current_device())
            buf = (torch.from_numpy(engine.export_packed()).to(dev) if rank == src
                   else torch.empty(n, dtype=torch.uint8, device=dev))
            dist.broadcast(buf, src=src)
            torch.cuda.synchronize()
            if rank != src:
                engine.import_packed(buf.cpu().numpy())
    else:
        n = engine.packed_size()
        buf = torch.from_numpy(engine.export_packed()) if rank == src else torch.empty(n, dtype=torch.uint8)
        dist.broadcast(buf, src=src)
        if rank != src:
            engine.import_packed(buf.numpy())


def max_over_ranks(value):
    """MAX all-reduce of a python float (the timing rule of bench.py)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
