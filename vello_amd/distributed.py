"""Multi-GPU sharding of independent scenes.

The path has no intra-frame partition worth xGMI traffic: frames/scenes are the independent units.  One
process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm), scene i -> rank i mod world, every rank
runs the full pipeline privately, and the ONE exchange step is the gather of finished RGBA8 frames
(w*h*4 bytes each) to rank 0.  With 7 point-to-point xGMI links per GPU each peer has its own link to the
root, so the gather is link-parallel; no all-reduce exists anywhere on this path.
"""
import torch
import torch.distributed as dist


def shard_scenes(n_scenes, rank, world):
    """Round-robin assignment: the scene indices rank `rank` of `world` renders."""
    return list(range(rank, n_scenes, world))


def gather_frames(frame, rank, world, dst=0, out=None):
    """Gathers one frame per rank to `dst`.  `frame`: uint8 tensor (H, W, 4) on this rank's device.
    Returns the list of `world` frames on dst (reusing `out` if given), None elsewhere."""
    if world == 1:
        return [frame]
    if rank == dst:
        if out is None:
            out = [torch.empty_like(frame) for _ in range(world)]
        dist.gather(frame, out, dst=dst)
        return out
    dist.gather(frame, None, dst=dst)
    return None
