#!/usr/bin/env python3
"""What of bench.py's N > 1 path a ONE-GPU box can execute on hardware: a process group on RCCL (backend "nccl", world size 1 as
torchrun launches it), the engine's frame handed to torch.distributed.gather / all_reduce(MAX) / barrier the way bench.py's exchange
and timing do it, against the oracle-checked frame of smoke().  Not a scaling measurement -- the collective has no peer -- but the
library, the stream order between the engine's streams and torch's, and the calls themselves are the ones N ranks make.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 scripts/rccl_world1_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import bench
import vello_amd


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    print("backend", dist.get_backend(), "world", world, "rccl/nccl version", torch.cuda.nccl.version())
    wl = bench.Workload("d2", 0)
    eng = vello_amd.Engine(device=local, capacities=wl.caps)
    eng.upload_scene(wl.packed, wl.layout)
    eng.set_frames_in_flight(4)
    w, h = wl.width, wl.height
    ring = [torch.zeros((h, w, 4), dtype=torch.uint8, device=f"cuda:{local}") for _ in range(4)]
    gathered = [torch.zeros_like(ring[0]) for _ in range(world)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for i in range(n):
        eng.render_resident(w, h, bench.BASE_COLOR, wl.aa, out=ring[i % 4])
        if i >= 3:
            eng.sync_frame(3)
            dist.gather(ring[(i - 3) % 4], gathered if rank == 0 else None, dst=0)   # (bench.py's exchange, called for every world size here)
    assert eng.sync() == 0, eng.bump()
    dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    t = torch.tensor([el], device=f"cuda:{local}", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    eng.render_resident(w, h, bench.BASE_COLOR, wl.aa, out=ring[0])
    eng.sync()
    dist.gather(ring[0], gathered if rank == 0 else None, dst=0)
    torch.cuda.synchronize()
    same = bool(torch.equal(gathered[0], ring[0]))
    print(f"{n} frames rendered with a gather per frame in {el * 1e3:.1f} ms ({n / el:.0f} frames/s incl. the first frames' warm-up); "
          f"all_reduce(MAX) of the time {float(t.item()) * 1e3:.1f} ms; gathered frame == rendered frame: {same}")
    dist.destroy_process_group()
    assert same


if __name__ == "__main__":
    main()
