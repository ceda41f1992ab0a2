"""Frame-parallel data parallelism: one process per GPU, frames sharded in contiguous blocks, no
data-path collective except ONE all-gather of the fixed-capacity 3D-keypoint tensor per batch
(RCCL over xGMI when the backend is "nccl"; gloo on CPU for tests).

The reference processes one frame at a time and has no collective on this path
(perception/pipeline.py:183); frames are independent, so this is pure
weak-scaling data parallelism with replicated weights.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def launched_by_torchrun():
    """True under `python -m torch.distributed.run` (any world size): the rendezvous variables are all there."""
    return all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, local_rank, world).
    A process group is created whenever the process was started by torchrun - also at world size 1, so that a single-GPU
    box exercises the same RCCL communicator set-up and the same collectives as the 8-GPU node; a plain `python` process
    (no rendezvous variables) stays without one and the helpers below pass through."""
    rank, local_rank, world = env_world()
    if (world > 1 or launched_by_torchrun()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, local_rank, world


def shard(total, rank, world):
    """Contiguous block of `total` frames owned by `rank`: (start, count).  Blocks differ by at most one frame."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def all_gather_keypoints(points, total_frames=None):
    """points: [frames_local, K, cap, 4] -> [total, K, cap, 4], ordered by rank, i.e. by global frame index for the
    contiguous blocks of `shard`.  ONE collective.  `all_gather_into_tensor` needs the same shape on every rank:
    with `total_frames` given (the global frame count the blocks were cut from with `shard`) and not divisible by the
    world size, every rank pads its block with NaN rows (NaN = unused slot, as inside the payload) to the largest block
    and the padding rows are dropped after the gather.  `total_frames` is REQUIRED when more than one rank takes part: a rank
    cannot know from its own block whether the others hold the same number of frames, and ranks that enter the collective
    with different shapes hang or corrupt the output on RCCL."""
    if not dist.is_initialized():
        return points
    world, rank = dist.get_world_size(), dist.get_rank()
    points = points.contiguous()
    if total_frames is None:
        if world > 1:
            raise ValueError("all_gather_keypoints: total_frames (the global frame count the blocks were cut from with shard()) "
                             "is required when world size > 1")
        total_frames = points.shape[0]
    counts = [shard(total_frames, r, world)[1] for r in range(world)]
    if points.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {points.shape[0]} frames, shard({total_frames}, {rank}, {world}) says {counts[rank]}; "
                         "pass total_frames for uneven blocks")
    biggest = max(counts)
    if points.shape[0] < biggest:
        pad = torch.full((biggest - points.shape[0],) + tuple(points.shape[1:]), float("nan"), dtype=points.dtype, device=points.device)
        points = torch.cat([points, pad])
    out = torch.empty((world * biggest,) + tuple(points.shape[1:]), dtype=points.dtype, device=points.device)
    dist.all_gather_into_tensor(out, points)
    if min(counts) == biggest:
        return out
    return torch.cat([out[r * biggest:r * biggest + c] for r, c in enumerate(counts)])


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
