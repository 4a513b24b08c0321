"""N > 1 path on CPU: two gloo ranks shard scenes round-robin, render (SIMT-emulated kernels stand in for
the GPU here: test infrastructure) and gather frames to rank 0, which checks every frame against the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE = 160
SEED0 = 0x5EED0001


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_scenes, result_path):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vello_amd._lib as L

    L._use_library(os.path.join(ROOT, "tests", "simt_emu", "libvello_emu.so"))
    import vello_amd
    import workloads
    from vello_amd.distributed import gather_frames, shard_scenes

    engine = vello_amd.Engine()
    mine = shard_scenes(n_scenes, rank, world)
    frames_at_root = {}
    for step in range((n_scenes + world - 1) // world):
        if step < len(mine):
            scene = workloads.paris_like_scene(SEED0 + mine[step], n_paths=120, size=float(SIZE))
            packed, layout = scene.resolve()
            img, bump = engine.render(packed, layout, SIZE, SIZE, 0xFFFFFFFF, vello_amd.AaConfig.Msaa16)
            assert bump["failed"] == 0
        else:
            img = np.zeros((SIZE, SIZE, 4), dtype=np.uint8)
        got = gather_frames(torch.from_numpy(img), rank, world)
        if rank == 0:
            for r, f in enumerate(got):
                ix = step * world + r
                if ix < n_scenes:
                    frames_at_root[ix] = f.numpy().copy()
    dist.barrier()
    if rank == 0:
        np.savez(result_path, **{f"f{k}": v for k, v in frames_at_root.items()})
    dist.destroy_process_group()


def test_two_rank_scene_sharding_and_gather(built, tmp_path):
    import workloads
    from oracle.oracle import Oracle
    from vello_amd.distributed import shard_scenes

    assert shard_scenes(5, 0, 2) == [0, 2, 4] and shard_scenes(5, 1, 2) == [1, 3]
    n_scenes, world = 3, 2
    result = str(tmp_path / "frames.npz")
    mp.spawn(_worker, args=(world, _free_port(), n_scenes, result), nprocs=world, join=True)
    frames = np.load(result)
    assert len(frames.files) == n_scenes
    o = Oracle()
    for i in range(n_scenes):
        packed, layout = workloads.paris_like_scene(SEED0 + i, n_paths=120, size=float(SIZE)).resolve()
        o.set_scene(packed, layout, SIZE, SIZE, 0xFFFFFFFF, 2)
        assert np.array_equal(frames[f"f{i}"], o.render()), f"scene {i} gathered at rank 0 differs from the oracle"


def test_frame_pipeline_exchanges_every_frame_once_in_order():
    # bench.py --gpus N keeps n frames in flight and gathers the oldest one while the younger ones render; the
    # bookkeeping is checked here with recording fakes (no GPU): order, wait ages, slot reuse only after the exchange
    from vello_amd.distributed import FramePipeline

    for n in (1, 2, 3, 6):
        for k in (1, 2, 5, 6, 7, 20):
            log, rendered, exchanged, handles = [], [], [], {}
            frame_of_slot = {}

            def render(slot):
                assert handles.get(slot) is None, "slot re-rendered before its exchange was waited for"
                frame_of_slot[slot] = len(rendered)
                rendered.append(slot)
                log.append(("render", slot))

            def wait_frame(age):
                assert 0 <= age < n and age < len(rendered)
                log.append(("wait", len(rendered) - 1 - age))

            def exchange(slot):
                f = frame_of_slot[slot]
                assert ("wait", f) in log, "exchange before the frame was waited for"
                exchanged.append(f)
                handles[slot] = f
                return slot

            def wait_exchange(slot):
                handles[slot] = None

            p = FramePipeline(n, render, wait_frame, exchange, wait_exchange)
            for _ in range(k):
                p.step()
            p.flush()
            assert exchanged == list(range(k)), (n, k, exchanged)
            assert p.issued == 0
    # single-GPU form: no exchange, nothing but renders into rotating slots
    calls = []
    p = FramePipeline(3, calls.append, lambda age: calls.append(("wait", age)))
    for _ in range(7):
        p.step()
    p.flush()
    assert calls == [0, 1, 2, 0, 1, 2, 0]
