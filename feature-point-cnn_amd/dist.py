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


class WeightBroadcastError(RuntimeError):
    """Raised on EVERY rank when the start-up exchange cannot go ahead (the source rank failed to load the
    checkpoint, or some rank's engine would lay the blob out differently)."""


def _coll_device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def broadcast_packed_weights(engine, state_dict, src=0):
    """Rank `src` loads `state_dict` into its engine; every other rank receives the packed, BN-folded blob.

    Three collectives, all of fixed size until every rank has agreed to the last one:
      1. a 4-word status from `src`: (loaded ok, blob bytes, launch-plan hash lo / hi).  If the source rank failed in
         load_state_dict (missing key, shape, FPC_E_RANGE) every rank raises WeightBroadcastError here, instead of the
         others waiting in a broadcast for the backend's time-out;
      2. a MAX all-reduce of "my engine disagrees" (blob size or plan hash: another dtype, arch, build or FPC_* plan
         knob) -- again every rank raises together;
      3. the blob.  It carries its own tag, which fpc_import_packed / fpc_mark_weights_loaded verify once more.
    A C/C++ host does the same through fpc_broadcast_weights(ctx, ncclComm_t, root) (include/fpc.h)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        engine.load_state_dict(state_dict)
        return
    rank = dist.get_rank()
    dev = _coll_device()
    err = None
    if rank == src:
        try:
            engine.load_state_dict(state_dict)
        except Exception as e:              # reported to every rank below, then re-raised here
            err = e
    n = engine.packed_size()
    ph = engine.plan_hash() if hasattr(engine, "plan_hash") else 0
    status = torch.tensor([0 if err is not None else 1, n, ph & 0xffffffff, ph >> 32], dtype=torch.int64, device=dev)
    dist.broadcast(status, src=src)
    ok, src_n, src_lo, src_hi = (int(v) for v in status.cpu())
    if not ok:
        if err is not None:
            raise WeightBroadcastError("rank %d could not load the checkpoint: %s" % (src, err)) from err
        raise WeightBroadcastError("rank %d could not load the checkpoint (see its log)" % src)
    mine_differs = src_n != n or (src_lo | (src_hi << 32)) != ph
    flag = torch.tensor([1 if mine_differs else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        raise WeightBroadcastError(
            "rank %d: packed-weight layout %s rank %d's (%d bytes, plan %016x here; %d bytes, plan %016x there): every "
            "rank must run the same build with the same dtype / arch / plan knobs"
            % (rank, "differs from" if mine_differs else "matches, but another rank's differs from", src, n, ph, src_n,
               src_lo | (src_hi << 32)))
    if dist.get_backend() == "nccl":
        target, finish = nccl_receive_target(engine, rank == src, n, dev)
        dist.broadcast(target, src=src)        # RCCL over xGMI, device to device (16 MB, start-up only)
        torch.cuda.synchronize()
        finish()
    else:
        buf = torch.from_numpy(engine.export_packed()) if rank == src else torch.empty(n, dtype=torch.uint8)
        dist.broadcast(buf, src=src)
        if rank != src:
            engine.import_packed(buf.numpy())


def nccl_receive_target(engine, is_src, n, dev):
    """(tensor, finish) for the blob collective of a device backend.  The tensor every rank hands to dist.broadcast is a
    VIEW OF THE LIBRARY'S OWN device blob (fpc_packed_device_ptr): the source sends from where fpc_load_weights packed,
    a receiver's bytes land where the kernels read them -- no staging buffer, no host round trip; `finish` then has the
    library verify the tag that arrived (fpc_mark_weights_loaded).  The three collectives in front of this one have
    already made every rank agree on the size and the plan hash, so every rank enters the same collective with the
    same byte count whichever target it uses.  If the view cannot be made on some rank (or FPC_DIST_ZERO_COPY=0), that
    rank receives into a torch-owned device buffer and hands it over with fpc_import_packed_device -- one device-to-
    device copy, still no host hop.  (Round 4's default went device -> host -> device on every receiver.)"""
    view = None
    if os.environ.get("FPC_DIST_ZERO_COPY", "1") != "0":
        try:
            view = engine.packed_view()
            if view.numel() != n or view.dtype != torch.uint8:
                view = None
        except Exception:
            view = None
    if view is not None:
        return view, (lambda: None) if is_src else engine.mark_weights_loaded
    buf = torch.empty(n, dtype=torch.uint8, device=dev)
    if is_src:
        buf.copy_(torch.from_numpy(engine.export_packed()))      # (the source's fallback only: its view could not be made)
        return buf, (lambda: None)
    return buf, (lambda: engine.import_packed_device(buf))


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
