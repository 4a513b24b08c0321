#!/usr/bin/env python
"""Benchmark of the hot path: frames/s on the paris-30k-like scene, 1600x1600, MSAA16 (BASELINE config C3).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the pipeline (pathtag scan ... fine) over one resident packed scene, producing
one RGBA8 frame in device memory.  With N > 1 every rank renders its own independent scene (seeds
0x5EED0001 + rank: weak scaling, no data-path collective) and each step ends with the ONE exchange the path
has: the gather of the finished frames to rank 0 over RCCL/xGMI.  The scene bytes are resident in HBM before
the timed region (PCIe-inclusive numbers are in DESIGN.md).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel: algorithmic bytes of that stage
(DESIGN.md "algorithmic bytes") / its mean launch duration measured with HIP events on the engine's stream
inside the timed region (with --in-flight > 1 the kernels of neighbouring frames share the CUs during that
launch; `avg_launch_ms_isolated`/`achieved_isolated` repeat the measurement with one frame at a time).
`cpu_baseline` times the CPU oracle (a port of the reference's CPU shaders +
fine.wgsl, single thread) on a bounded sample of the same workload on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Frames in flight run on one HIP stream each; the ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4) and streams that share a queue serialise (measured: 4 frames in flight on 4 queues = 2229
# frames/s, 3 = 2769, 6 on 8 queues = 3040).  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

WIDTH = HEIGHT = 1600
BASE_COLOR = 0xFFFFFFFF
SEED0 = 0x5EED0001
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def stage_bytes(layout, bump, n_tag_words, scene_len, width, height, ptcl_words):
    """Algorithmic HBM bytes per stage for one frame (each buffer written once, read once per consumer;
    SURVEY.md 8d d4, split per kernel in DESIGN.md)."""
    P, D = layout.n_paths, layout.n_draw_objects
    L, A, C, G, Bd = bump["lines"], bump["tile"], bump["seg_counts"], bump["segments"], bump["binning"]
    K = layout.n_clips
    Tw = n_tag_words
    path_data = (layout.draw_tag_base - layout.path_data_base) * 4
    return {
        "pathtag_scan": 4 * Tw + 20 * Tw + 16 * P,
        "flatten": (4 * Tw + 20 * Tw + path_data) + 24 * L + 24 * P,   # single pass: tags, monoids, path data read once
        "draw_scan": 4 * D + 24 * D + 16 * D + 4 * D,
        "clip": 8 * K + 24 * K + 16 * K + 16 * K,
        "binning": 16 * D + 24 * D + 16 * D + 4 * Bd,
        "tile_alloc": 4 * D + 16 * D + 32 * D + 8 * A,
        "path_count": 2 * 24 * L + 16 * C + 8 * C,   # lines read twice, tile RMW per crossing, SegmentCount written
        "backdrop": 32 * D + 16 * A,
        "coarse": 4 * Bd + 2 * 8 * A + 4 * ptcl_words + 4 * A,
        "path_tiling": 8 * C + 24 * C + 8 * C + 24 * G,
        "fine": 4 * ptcl_words + 24 * G + 4 * width * height,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--in-flight", type=int, default=6,
                    help="frames the engine keeps in flight (wgpu queues recordings the same way); 1 = serial frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true",
                    help="skip the serial / PCIe / CPU passes (for rocprofv3 runs: every launch it sees is then a "
                         "warm-up or timed-region launch)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (vello_amd has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist

        dist.init_process_group("nccl")  # RCCL

    import vello_amd
    import workloads

    scene = workloads.paris_like_scene(SEED0 + rank)
    packed, layout = scene.resolve()
    n_tag_words = layout.path_data_base - layout.path_tag_base
    engine = vello_amd.Engine(device=local_rank)
    engine.upload_scene(packed, layout)
    aa = vello_amd.AaConfig.Msaa16
    nif = max(1, min(args.in_flight, 8))
    engine.set_frames_in_flight(nif)
    # every in-flight frame owns its target (a swapchain of nif images)
    ring = [torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.uint8, device=f"cuda:{local_rank}") for _ in range(nif)]
    frame = ring[0]
    gathered = [torch.zeros_like(frame) for _ in range(world)] if (distributed and rank == 0) else None

    from vello_amd.distributed import gather_frames

    from vello_amd.distributed import FramePipeline

    def exchange(slot):
        gather_frames(ring[slot], rank, world, dst=0, out=gathered)
        ev = torch.cuda.Event()
        ev.record()
        return ev

    pipe = FramePipeline(
        nif,
        render=lambda slot: engine.render_resident(WIDTH, HEIGHT, BASE_COLOR, aa, out=ring[slot]),
        wait_frame=engine.sync_frame,                      # the frame is complete before its collective goes on torch's stream
        exchange=exchange if distributed else None,
        wait_exchange=lambda ev: ev.synchronize(),         # a slot is re-rendered only after its collective has read it
    )
    step, flush = pipe.step, pipe.flush

    # warmup, with every stage under HIP events to find the dominant kernel
    engine.set_profiling(vello_amd.renderer.STAGES)
    for _ in range(max(args.warmup, 1)):
        step()
    flush()
    torch.cuda.synchronize()
    rc = engine.sync()
    if rc != 0:
        raise SystemExit(f"frame failed: {rc} {engine.bump()}")
    warm_ms = engine.stage_ms()
    bump = engine.bump()
    dominant = max(warm_ms, key=lambda k: warm_ms[k][0] / max(warm_ms[k][1], 1))

    # timed region: exactly K steps, events only around the dominant kernel
    engine.set_profiling([dominant])
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    flush()
    rc = engine.sync()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if rc != 0:
        raise SystemExit(f"frame failed in the timed region: {rc} {engine.bump()}")
    dom_ms, dom_n = engine.stage_ms()[dominant]
    if distributed:
        t = torch.tensor([elapsed], device=f"cuda:{local_rank}", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the exchange step on its own ("report the gather time separately"): nothing else on the GPUs
    exchange_ms = None
    if distributed:
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(10):
            gather_frames(ring[0], rank, world, dst=0, out=gathered)
        torch.cuda.synchronize()
        dist.barrier()
        exchange_ms = (time.perf_counter() - t1) / 10 * 1e3

    # serial-frame passes (separate, not part of `value`): one frame at a time gives the frame latency and
    # the isolated per-kernel durations (no other frame's kernels sharing the CUs)
    n_serial = 0 if args.timed_only else min(args.steps, 50)
    engine.set_profiling([])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n_serial):
        engine.render_resident(WIDTH, HEIGHT, BASE_COLOR, aa, out=frame)
        engine.sync_frame(0)
    serial_ms = (time.perf_counter() - t1) / max(n_serial, 1) * 1e3
    engine.set_profiling(vello_amd.renderer.STAGES)
    for _ in range(n_serial):
        engine.render_resident(WIDTH, HEIGHT, BASE_COLOR, aa, out=frame)
        engine.sync_frame(0)
    engine.sync()
    all_ms = engine.stage_ms()
    engine.set_profiling([])

    # PCIe-inclusive rates (never `value`): the packed scene starts in host memory every frame.
    #  (a) serial: upload, render, wait -- what a blocking render_to_texture does;
    #  (b) pipelined: vello_hip_render_frame puts each frame's scene into the next in-flight slot, so the H2D copy of
    #      frame i+1 overlaps the kernels of the frames before it (animation form).
    n_pcie = 0 if args.timed_only else 20
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n_pcie):
        engine.upload_scene(packed, layout)
        engine.render_resident(WIDTH, HEIGHT, BASE_COLOR, aa, out=frame)
        engine.sync()
    pcie_fps = n_pcie / max(time.perf_counter() - t1, 1e-9)
    n_pipe = 0 if args.timed_only else 60
    for i in range(min(n_pipe, nif)):  # warm the private scene slots (first use allocates)
        engine.render_frame(packed, layout, WIDTH, HEIGHT, BASE_COLOR, aa, out=ring[i % nif])
    engine.sync()
    t1 = time.perf_counter()
    for i in range(n_pipe):
        engine.render_frame(packed, layout, WIDTH, HEIGHT, BASE_COLOR, aa, out=ring[i % nif])
    engine.sync()
    pcie_pipelined_fps = n_pipe / max(time.perf_counter() - t1, 1e-9)
    engine.upload_scene(packed, layout)  # back to the shared resident scene
    if args.timed_only:
        all_ms = {k: (0.0, 0) for k in vello_amd.renderer.STAGES}
        all_ms[dominant] = (dom_ms, dom_n)

    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    ptcl_words = 64 * ((WIDTH + 15) // 16) * ((HEIGHT + 15) // 16) + bump["ptcl"]
    sb = stage_bytes(layout, bump, n_tag_words, packed.nbytes, WIDTH, HEIGHT, ptcl_words)
    dom_avg_s = (dom_ms / max(dom_n, 1)) * 1e-3
    achieved = sb[dominant] / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
    frame_bytes = sum(sb.values())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.steps / elapsed

    # HBM traffic of the dominant kernel from the PMC passes (collected separately, as the pool requires, by
    # scripts/gpu_calib.sh; corrected with the factors calibrated in the same session)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            traffic = json.load(fh)["kernels"].get(f"k_{dominant}", {}).get("traffic_bytes")
    except (OSError, ValueError):
        pass

    result = {
        "metric": "frames/sec paris-30k 1600x1600 MSAA16; scenes/sec at 1/2/4/8 GPU",
        "value": round(value, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32+u32",
        "data": "synthetic",
        "config": {
            "workload": "paris-30k-like SYNTHETIC scene (real paris-30k.svg is not in the reference tree), "
                        f"seed 0x{SEED0:X}+rank, {layout.n_paths} paths, {n_tag_words * 4} path tags, "
                        f"{packed.nbytes / 1e6:.2f} MB packed encoding, {WIDTH}x{HEIGHT}, MSAA16",
            "baseline_config": "configs[2]",
            "parallelism": f"scenes{world}" if distributed else "single",
            "exchange": "RCCL gather of RGBA8 frames to rank 0 each step" if distributed else "none",
            "exchange_alone_ms": None if exchange_ms is None else round(exchange_ms, 4),
            "bump": bump,
            "frames_in_flight": nif,
            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "serial_frame_latency_ms": round(serial_ms, 4),
            "pcie_inclusive_frames_per_s": round(pcie_fps, 2),
            "pcie_inclusive_pipelined_frames_per_s": round(pcie_pipelined_fps, 2),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": f"k_{dominant}",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "peak_measured": 6290.0,  # float4 copy on this part (MI355X_MICROARCH.md), for reference
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            "traffic_source": "profiles/pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per launch)",
            "algorithmic_bytes_per_launch": int(sb[dominant]),
            "avg_launch_ms": round(dom_avg_s * 1e3, 5),
            "avg_launch_ms_isolated": round(all_ms[dominant][0] / max(all_ms[dominant][1], 1), 5),
            "achieved_isolated": round(sb[dominant] / (all_ms[dominant][0] / max(all_ms[dominant][1], 1) * 1e-3) / 1e9, 2),
            "frame_algorithmic_bytes": int(frame_bytes),
            "frame_achieved_GBps": round(frame_bytes / (elapsed / args.steps) / 1e9 * (1 if not distributed else 1), 2),
            "stage_ms": {k: round(v[0] / max(v[1], 1), 5) for k, v in all_ms.items()},
        },
    }

    if world == 1 and not args.no_cpu_baseline and not args.timed_only:
        from oracle.oracle import Oracle

        o = Oracle()
        o.set_scene(packed, layout, WIDTH, HEIGHT, BASE_COLOR, int(aa))
        o.render()  # warm
        n_cpu = 12  # ~10 s of single-thread CPU work + ~3 s with the fine stage threaded
        t0 = time.perf_counter()
        for _ in range(n_cpu):
            ref = o.render()
        cpu_s = (time.perf_counter() - t0) / n_cpu
        same = bool(np.array_equal(ref, frame.cpu().numpy()))
        # the same oracle with its tile-parallel fine stage on up to 64 threads (the other stages stay serial)
        n_thr = max(1, min(os.cpu_count() or 1, 64))
        o.set_threads(n_thr)
        o.render()
        t0 = time.perf_counter()
        for _ in range(n_cpu):
            o.render()
        cpu_mt_s = (time.perf_counter() - t0) / n_cpu
        result["cpu_baseline"] = {
            "value": round(1.0 / cpu_s, 4),
            "unit": "frames/s",
            "cores": 1,
            "value_fine_threaded": round(1.0 / cpu_mt_s, 4),
            "cores_fine_threaded": n_thr,
            "kind": "port",
            "sample": f"{n_cpu} full frames of the same scene (C restatement of vello_shaders/src/cpu + fine.wgsl, "
                      f"single thread, {os.cpu_count()} host cores present); output identical to GPU frame: {same}",
        }
    print(json.dumps(result))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
