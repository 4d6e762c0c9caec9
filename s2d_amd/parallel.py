"""One process per GPU; clips are the independent units of the hot path (no cross-clip op in forward or loss,
SURVEY.md 8e), so ranks shard clips statically and the data path has no collective.  torch.distributed (backend
"nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests) is used only for the barrier and max-over-ranks timing."""
import os

import torch


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    import torch.distributed as dist
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return world, rank, local


def shard_clips(n_clips_global, rank, world):
    """static, contiguous, balanced shard of global clip ids for this rank (IMS_PER_BATCH / world clips per rank,
    data_video/build.py:21-35)"""
    per, rem = divmod(n_clips_global, world)
    start = rank * per + min(rank, rem)
    return list(range(start, start + per + (1 if rank < rem else 0)))


def fence(device=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(x, device="cpu"):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)


def whole_job_rate(units_per_rank, steps, elapsed_max, world):
    """value = units all ranks processed / max-over-ranks time"""
    return units_per_rank * world * steps / elapsed_max
