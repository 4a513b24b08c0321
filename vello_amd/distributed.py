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


class FramePipeline:
    """Bookkeeping of a ring of `n_in_flight` frames: render frame i into slot i % n, and once a frame is the oldest one in
    flight wait for it alone and hand it to `exchange` (the gather), so that the collective of frame i - (n - 1) overlaps
    the rendering of the n - 1 younger frames.  The callables keep it testable without a GPU:

      render(slot)            enqueue one frame into ring slot `slot`
      wait_frame(age)         block until the frame enqueued `age` render() calls ago is complete (0 = newest)
      exchange(slot)          the one exchange step of the path for the finished frame in `slot`; may return a handle
      wait_exchange(handle)   block until that exchange has finished reading the slot (called before the slot is reused)
    """

    def __init__(self, n_in_flight, render, wait_frame, exchange=None, wait_exchange=None):
        self.n = max(1, int(n_in_flight))
        self.render, self.wait_frame = render, wait_frame
        self.exchange, self.wait_exchange = exchange, wait_exchange
        self.issued = 0
        self.pending = [None] * self.n

    def _exchange(self, slot):
        if self.exchange is not None:
            self.pending[slot] = self.exchange(slot)

    def step(self):
        i = self.issued
        self.issued = i + 1
        slot = i % self.n
        if self.pending[slot] is not None:
            if self.wait_exchange is not None:
                self.wait_exchange(self.pending[slot])
            self.pending[slot] = None
        self.render(slot)
        if self.exchange is not None and i >= self.n - 1:
            self.wait_frame(self.n - 1)
            self._exchange((i - (self.n - 1)) % self.n)

    def flush(self):
        """Exchanges the frames still in flight (every rendered frame is exchanged exactly once) and restarts the count."""
        if self.exchange is not None:
            i = self.issued
            for age in range(min(self.n - 1, i) - 1, -1, -1):
                self.wait_frame(age)
                self._exchange((i - 1 - age) % self.n)
        self.issued = 0
