"""Multi-GPU sharding: one process per GPU, channels partitioned contiguously, no collective on the data path.

The demodulator path shards embarrassingly by channel (the reference even runs its two stereo burst
channels as two unrelated objects, JAERO/audioburstoqpskdemodulator.cpp:8-10), so the steady state has NO exchange
step.  torch.distributed (backend "nccl" = RCCL over xGMI on GPUs, "gloo" on CPU for tests) is used only for the two
edge operations the north star names:
  * fan_out_pcm : rank `src` holds one block of interleaved frames [nsamples, nch_total] and scatters each rank's
                  contiguous channel slice to it;
  * gather_softbits : fixed-size per-channel soft-bit slots (+counts) are gathered on rank `dst`.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple


def shard_range(nch_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous channel range [lo, hi) of `rank`: ch in [rank*N/W, (rank+1)*N/W)."""
    lo = (rank * nch_total) // world
    hi = ((rank + 1) * nch_total) // world
    return lo, hi


def init_from_env(backend: Optional[str] = None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* (as torchrun sets them); returns
    (rank, world, local_rank).  With WORLD_SIZE unset or 1 nothing is initialised."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("JAERO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() and not ranks_share_a_device() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def local_world_size() -> Optional[int]:
    """Ranks on this node, from whichever launcher exported it: torchrun (LOCAL_WORLD_SIZE), Slurm (SLURM_NTASKS_PER_NODE, possibly
    "8(x2)"), Open MPI (OMPI_COMM_WORLD_LOCAL_SIZE), MPICH / Intel MPI (MPI_LOCALNRANKS).  None when nobody said."""
    for key in ("LOCAL_WORLD_SIZE", "SLURM_NTASKS_PER_NODE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS"):
        v = os.environ.get(key)
        if v:
            digits = ""
            for ch in v.strip():
                if not ch.isdigit():
                    break
                digits += ch
            if digits:
                return int(digits)
    return None


def ranks_share_a_device() -> bool:
    """True when this node has fewer GPUs than local ranks (a multi-rank run squeezed onto a smaller lease: every rank then takes device
    local_rank % device_count, and the control plane runs over gloo because RCCL refuses two ranks on one GPU).

    The answer picks the backend (init_from_env), so it must be the SAME on every rank of the job: it is derived only from values every
    rank of a node sees alike -- the launcher's local world size where one is exported, otherwise the global WORLD_SIZE (conservative: a
    multi-node job of an unknown launcher with more ranks than one node has GPUs runs its control plane over gloo; JAERO_DIST_BACKEND
    overrides).  A per-rank guess from LOCAL_RANK (round 4) made rank 0 pick nccl and rank 1 gloo on a 2-rank / 1-GPU srun."""
    import torch

    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n == 0:
        return False
    lws = local_world_size()
    # one visible device per task under a per-task binding (srun --gpus-per-task=1 / --gpu-bind, or a launcher that sets ROCR_/HIP_VISIBLE_DEVICES per
    # rank): every rank owns the one GPU it sees, although the node runs several tasks.  The variables are the launcher's, the same on
    # every rank of the job, so the answer still is.
    if n == 1 and (os.environ.get("SLURM_GPUS_PER_TASK") == "1" or os.environ.get("SLURM_TRES_PER_TASK", "").replace("gres/", "").startswith("gpu:1")
                   or os.environ.get("JAERO_ONE_GPU_PER_RANK") == "1"):
        return False
    if lws is not None:
        return lws > n
    return int(os.environ.get("WORLD_SIZE", "1")) > n


def device_index(local: int) -> int:
    """The GPU of local rank `local`: its own one, or local % device_count when ranks share devices."""
    import torch

    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    return local % n if n else 0


def fan_out_pcm(frames, nch_total: int, nsamples: int, src: int = 0, device=None):
    """frames: int16 tensor [nsamples, nch_total] on rank `src` (ignored elsewhere).  Returns this rank's
    [nsamples, hi-lo] slice (contiguous, on `device`)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return frames
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(nch_total, rank, world)
    if device is None:
        device = frames.device if frames is not None else torch.device("cpu")
    # gloo's point-to-point path takes host tensors only: staged through the host there (the control plane of ranks that share a device)
    via_host = dist.get_backend() == "gloo" and torch.device(device).type != "cpu"
    wire = torch.device("cpu") if via_host else device
    mine = torch.empty((nsamples, hi - lo), dtype=torch.int16, device=wire)
    # point-to-point (xGMI is point-to-point; the volumes are tiny next to one link): src sends each peer its slice
    if rank == src:
        reqs = []
        for r in range(world):
            l, h = shard_range(nch_total, r, world)
            part = frames[:, l:h].contiguous()
            if r == src:
                mine.copy_(part)
            else:
                reqs.append(dist.isend(part.to(wire), dst=r))
        for q in reqs:
            q.wait()
    else:
        dist.recv(mine, src=src)
    return mine.to(device) if via_host else mine


def gather_softbits(soft, counts, nch_total: int, dst: int = 0):
    """soft: int16 [nch_local, cap], counts: int32 [nch_local] (same cap on every rank).  Returns on `dst`
    (soft_all [nch_total, cap], counts_all [nch_total]); (None, None) elsewhere."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return soft, counts
    rank, world = dist.get_rank(), dist.get_world_size()
    cap = soft.shape[1]
    via_host = dist.get_backend() == "gloo" and soft.device.type != "cpu"  # gloo moves host tensors only
    wire = torch.device("cpu") if via_host else soft.device
    if rank == dst:
        soft_all = torch.empty((nch_total, cap), dtype=soft.dtype, device=wire)
        counts_all = torch.empty((nch_total,), dtype=counts.dtype, device=wire)
        for r in range(world):
            l, h = shard_range(nch_total, r, world)
            if r == dst:
                soft_all[l:h].copy_(soft)
                counts_all[l:h].copy_(counts)
            else:
                dist.recv(soft_all[l:h], src=r)
                dist.recv(counts_all[l:h], src=r)
        return (soft_all.to(soft.device), counts_all.to(counts.device)) if via_host else (soft_all, counts_all)
    dist.send(soft.contiguous().to(wire), dst=dst)
    dist.send(counts.contiguous().to(wire), dst=dst)
    return None, None
