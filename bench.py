#!/usr/bin/env python
"""Benchmark of the hot path: frames/s on paris-30k-like scenes, 1600x1600, MSAA16 (BASELINE config C3).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the pipeline (pathtag scan ... fine) over one resident packed scene, producing
one RGBA8 frame in device memory.  With N > 1 every rank renders its own independent scene (seeds
0x5EED0001 + rank: weak scaling, no data-path collective) and each step ends with the ONE exchange the path
has: the gather of the finished frames to rank 0 over RCCL/xGMI.  The scene bytes are resident in HBM before
the timed region (the PCIe-inclusive rates are reported in `config`, never as `value`).

Workloads (both SYNTHETIC: the real paris-30k.svg is not in the reference tree):
  d2      the scene SURVEY.md 8d d2 fixes for C3: 70 % stroked polylines / 25 % polygons / 5 % cubic blobs, steps
          4-40 px (workloads.paris_like_scene_d2).  It needs pools beyond the reference's fixed sizes (D2_CAPS).
          This is the workload of `value`.
  r1mix   round 1's stroke-light mix sized for the reference's own pools (workloads.paris_like_scene), kept beside
          it in config.secondary for continuity with BENCH_r01.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel: SURVEY 8d d4's per-stage terms evaluated with
the frame's own bump counters / its launch duration measured with HIP events on the engine's stream, ONE FRAME AT
A TIME (isolated: `frac`), and with --in-flight frames sharing the CUs (`frac_overlapped`, context only).
`frame_algorithmic_bytes` is d4's single formula.  `cpu_baseline` times the CPU oracle (a C port of the reference's
CPU shaders + fine.wgsl) on a bounded sample of the same workload on rank 0 at N=1.
`config.other_configs` carries BASELINE's other single-GPU configurations (C2 Tiger 1024^2 MSAA8, C4 mmark-50k 2048^2
MSAA16; bounded, ~1 s each); `config.value_200_steps` is the headline once more over a fixed 200 steps (SURVEY 8d d1's
"median of >= 200 frames" whatever --steps the caller passed).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Frames in flight run on one HIP stream each; the ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4) and streams that share a queue serialise.  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

WIDTH = HEIGHT = 1600
BASE_COLOR = 0xFFFFFFFF
SEED0 = 0x5EED0001
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
# pools for the d2 scene (elements; vello_hip_capacities).  Its demand: 3.06 M lines, 4.28 M crossings / segments,
# 9.98 M path tiles, 7.35 M dynamic PTCL words -- beyond config.rs:401-408's 2^21 / 2^21 / 2^21 / 2^23.
THREADED_STAGES = "flatten (per tag), path_count (per line), coarse (per bin), path_tiling (per crossing), fine (per tile); the scans, binning, tile_alloc and backdrop (2 % of the single-thread time) stay serial"
D2_CAPS = {"lines": 1 << 22, "tiles": 1 << 24, "seg_counts": 1 << 23, "segments": 1 << 23, "ptcl": 3 << 22}


def stage_bytes(layout, bump, n_tag_words, width, height, ptcl_words):
    """Algorithmic HBM bytes per stage for one frame: the terms of SURVEY.md 8d d4 assigned to the kernel that moves
    them (each buffer written once, read once per consumer); the sum differs from d4's single figure only by the
    packed scene (S vs the parts the stages read) and is printed separately."""
    P, D = layout.n_paths, layout.n_draw_objects
    L, A, C, G, Bd = bump["lines"], bump["tile"], bump["seg_counts"], bump["segments"], bump["binning"]
    K = layout.n_clips
    Tw = n_tag_words
    path_data = (layout.draw_tag_base - layout.path_data_base) * 4
    return {
        "pathtag_scan": 4 * Tw + 20 * Tw + 24 * P,                      # tags r, monoids w, bbox clear
        "flatten": 4 * Tw + 20 * Tw + path_data + 24 * L + 24 * P,      # tags, monoids, path data r; lines w; bbox RMW
        "draw_scan": 4 * D + 16 * D + 8 * D + 24 * P,                   # tags r, monoids w, info w, bbox r
        "clip": 64 * K,
        "binning": 16 * D + 24 * P + 16 * D + 4 * D + 4 * Bd,           # monoids, bbox r; draw_bbox w; tags; bin_data w
        "tile_alloc": 4 * D + 16 * D + 32 * D + 8 * A,                  # tags, draw_bbox r; Path w; tiles zeroed
        "path_count": 24 * L + 16 * C + 8 * C,                          # lines r; tile RMW per crossing; SegmentCount w
        "backdrop": 32 * D + 16 * A,                                    # Path r; tile backdrop r + w
        "coarse": 4 * Bd + 16 * D + 32 * D + 8 * A + 4 * A + 4 * ptcl_words,   # bin_data, monoids, Path, tiles r; seg_ix w; PTCL w
        "path_tiling": 8 * C + 24 * C + 8 * C + 24 * G,                 # SegmentCount r; line gather; tile r; segments w
        "fine": 4 * ptcl_words + 24 * G + 4 * width * height,           # PTCL r, segments r, RGBA8 w
    }


def d4_frame_bytes(scene_len, layout, bump, n_tag_words, width, height, ptcl_words):
    """SURVEY.md 8d d4, verbatim: bytes = S + 48 Tw + 96 P + 168 D + 8 Bd + 48 L + 36 A + 48 C + 48 G + 8 Wp + 4 Npx."""
    return (scene_len + 48 * n_tag_words + 96 * layout.n_paths + 168 * layout.n_draw_objects + 8 * bump["binning"] + 48 * bump["lines"]
            + 36 * bump["tile"] + 48 * bump["seg_counts"] + 48 * bump["segments"] + 8 * ptcl_words + 4 * width * height)


def pct(xs, q):
    xs = sorted(xs)
    if not xs:
        return None
    return xs[min(len(xs) - 1, max(0, int(round(q * (len(xs) - 1)))))]


def measure_copy_peak(local_rank):
    """Device-to-device float4 copy of 1 GiB on this GPU, this run (scripts/calib/copy_bw.hip, the kernel shape
    MI355X_MICROARCH.md quotes 6.29 TB/s for): bytes read + written / best of 5, GB/s.  None if the helper is not built."""
    import ctypes

    path = os.path.join(ROOT, "scripts", "calib", "libcopy_bw.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    lib.copy_bw_gbps.restype = ctypes.c_double
    lib.copy_bw_gbps.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_int]
    g = lib.copy_bw_gbps(local_rank, 1 << 30, 5)
    return g if g > 0 else None


def git_head():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        pass
    try:  # the GPU box gets a snapshot without .git: scripts/grun.sh leaves the stamp beside it
        return open(os.path.join(ROOT, ".commit_stamp")).read().strip() or None
    except OSError:
        return None


class Workload:
    def __init__(self, key, rank):
        import vello_amd
        import workloads

        self.key = key
        self.width = self.height = WIDTH
        self.aa = vello_amd.AaConfig.Msaa16
        self.caps = None
        if key == "d2":
            self.scene = workloads.paris_like_scene_d2(SEED0 + rank)
            self.caps = dict(D2_CAPS)
            self.mix = "70 % stroked open polylines (width 0.5-4 px, 8-60 vertices, step 4-40 px) / 25 % filled polygons / 5 % cubic blobs (SURVEY 8d d2)"
            self.packed, self.layout = self.scene.resolve()
        elif key == "r1mix":
            self.scene = workloads.paris_like_scene(SEED0 + rank)
            self.mix = "16 % stroked polylines (width 0.5-3 px, step 1.5-7 px) / 81 % filled polygons / 3 % cubic blobs (round 1's mix, fits config.rs:401-408)"
            self.packed, self.layout = self.scene.resolve()
        elif key == "tiger":  # BASELINE configs[1]
            d = np.load(os.path.join(ROOT, "tests", "golden", "tiger_scene.npz"))
            self.packed, self.layout = d["packed"], vello_amd.Layout(*[int(v) for v in d["layout"]])
            self.width = self.height = 1024
            self.aa = vello_amd.AaConfig.Msaa8
            self.mix = "Ghostscript_Tiger.svg through the pico_svg-equivalent loader (tests/golden/tiger_scene.npz), fills + strokes"
        elif key == "mmark":  # BASELINE configs[3]
            self.scene = workloads.mmark_scene()
            self.packed, self.layout = self.scene.resolve()
            self.width = self.height = 2048
            self.mix = "mmark-style 50 000 stroked line / quad / cubic elements (examples/scenes/src/mmark.rs:43-202, PCG seed 0x5EED0002, no text label)"
        else:
            raise ValueError(key)
        self.n_tag_words = self.layout.path_data_base - self.layout.path_tag_base

    def describe(self, engine):
        caps = engine.capacities()
        return (f"paris-30k-like SYNTHETIC scene '{self.key}' (real paris-30k.svg is not in the reference tree), seed 0x{SEED0:X}+rank, "
                f"{self.layout.n_paths} paths: {self.mix}; {self.n_tag_words * 4} path tags, {self.packed.nbytes / 1e6:.2f} MB packed encoding, "
                f"{WIDTH}x{HEIGHT}, MSAA16; pools (elements): " + ", ".join(f"{k} {v}" for k, v in caps.items()))


def dominant_of(engine, stage_ms, kernel_ms):
    """(kernel name, its stage, its mean isolated launch ms, the stage's ms) of the longest launch among per-stage / per-kernel
    (ms, n) pairs measured one frame at a time."""
    per_kernel = {f"k_{st}": (stage_ms[st][0] / max(stage_ms[st][1], 1), st) for st in stage_ms if st not in engine.KERNELS}
    for st, names in engine.KERNELS.items():
        for kn in names:
            per_kernel[kn] = (kernel_ms[kn][0] / max(kernel_ms[kn][1], 1), st)
    k = max(per_kernel, key=lambda n: per_kernel[n][0])
    st = per_kernel[k][1]
    return k, st, per_kernel[k][0], stage_ms[st][0] / max(stage_ms[st][1], 1)


def measure_other_config(key, label, local_rank, budget_s=0.8):
    """One of BASELINE's other single-GPU configurations, bounded (about a second of GPU time): frames/s with 4 frames in
    flight, one-frame latency (median), the dominant kernel with its algorithmic bytes and HBM fraction.  Not the bench metric."""
    import vello_amd

    wl = Workload(key, 0)
    dev = f"cuda:{local_rank}"
    eng = vello_amd.Engine(device=local_rank, capacities=wl.caps)
    eng.upload_scene(wl.packed, wl.layout)
    w, h, aa = wl.width, wl.height, wl.aa
    nif = 4
    eng.set_frames_in_flight(nif)
    ring = [torch.zeros((h, w, 4), dtype=torch.uint8, device=dev) for _ in range(nif)]
    torch.cuda.synchronize()
    for i in range(12):
        eng.render_resident(w, h, BASE_COLOR, aa, out=ring[i % nif])
    if eng.sync() != 0:
        return {"config": label, "error": f"frame failed: {eng.bump()}"}
    t0 = time.perf_counter()
    n = 0
    while n < 40 or (time.perf_counter() - t0 < budget_s * 0.5 and n < 4000):
        eng.render_resident(w, h, BASE_COLOR, aa, out=ring[n % nif])
        n += 1
    eng.sync()
    fps = n / (time.perf_counter() - t0)
    eng.set_frames_in_flight(1)
    lat = []
    for _ in range(5):
        eng.render_resident(w, h, BASE_COLOR, aa, out=ring[0])
        eng.sync_frame(0)
    t0 = time.perf_counter()
    while len(lat) < 30 or (time.perf_counter() - t0 < budget_s * 0.3 and len(lat) < 400):
        t1 = time.perf_counter()
        eng.render_resident(w, h, BASE_COLOR, aa, out=ring[0])
        eng.sync_frame(0)
        lat.append((time.perf_counter() - t1) * 1e3)
    eng.set_profiling(vello_amd.renderer.STAGES)
    eng.stage_ms(); eng.kernel_ms()
    for _ in range(20):
        eng.render_resident(w, h, BASE_COLOR, aa, out=ring[0])
        eng.sync_frame(0)
    eng.sync()
    st_ms, k_ms = eng.stage_ms(), eng.kernel_ms()
    eng.set_profiling([])
    bump = eng.bump()
    kname, stage, k_iso, st_iso = dominant_of(eng, st_ms, k_ms)
    ptcl_words = 64 * ((w + 15) // 16) * ((h + 15) // 16) + bump["ptcl"]
    sb = stage_bytes(wl.layout, bump, wl.n_tag_words, w, h, ptcl_words)
    multi = stage in eng.KERNELS
    t_ms = st_iso if multi else k_iso
    return {
        "config": label, "scene": wl.mix, "size": [w, h], "aa": "msaa8" if int(aa) == 1 else ("msaa16" if int(aa) == 2 else "area"),
        "value": round(fps, 1), "unit": "frames/s", "frames_in_flight": nif, "timed_frames": n,
        "one_frame_latency_ms": round(pct(lat, 0.5), 4), "value_one_frame_at_a_time": round(1e3 / pct(lat, 0.5), 1),
        "dominant_kernel": kname, "dominant_kernel_ms": round(k_iso, 5),
        "algorithmic_bytes": int(sb[stage]), "bytes_and_time_of": "stage" if multi else "kernel",
        "frac": round(sb[stage] / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if t_ms > 0 else None,
        "stage_ms": {k: round(v[0] / max(v[1], 1), 5) for k, v in st_ms.items() if v[0] / max(v[1], 1) >= 0.002},
        "bump": bump,
    }


def run_workload(wl, args, rank, local_rank, world, timed_headline):
    """Uploads the scene and measures it.  For the headline workload the timed region is the contract's: exactly
    args.steps steps between barrier + synchronize on both sides."""
    import vello_amd
    from vello_amd.distributed import FramePipeline, gather_frames

    distributed = world > 1
    if distributed:
        import torch.distributed as dist
    dev = f"cuda:{local_rank}"
    engine = vello_amd.Engine(device=local_rank, capacities=wl.caps)
    engine.upload_scene(wl.packed, wl.layout)
    aa = vello_amd.AaConfig.Msaa16
    nif = max(1, min(args.in_flight, 8))
    engine.set_frames_in_flight(nif)
    ring = [torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.uint8, device=dev) for _ in range(nif)]
    frame = ring[0]
    gathered = [torch.zeros_like(frame) for _ in range(world)] if (distributed and rank == 0) else None
    torch.cuda.synchronize()  # torch's zero fills run on torch's stream, the frames on the engine's own
    steps = args.steps if timed_headline else max(20, args.steps // 2)

    def exchange(slot):
        gather_frames(ring[slot], rank, world, dst=0, out=gathered)
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def render(slot):
        engine.render_resident(WIDTH, HEIGHT, BASE_COLOR, aa, out=ring[slot])

    step_done = []  # host clock when the oldest frame in flight has completed (one entry per step once the ring is full)

    def local_done(slot):  # N = 1: no exchange, but the same back-pressure -- wait for the oldest frame before the next one
        step_done.append(time.perf_counter())
        return None

    pipe = FramePipeline(nif, render=render, wait_frame=engine.sync_frame, exchange=exchange if distributed else local_done,
                         wait_exchange=lambda ev: ev.synchronize())

    # warm-up, with every stage under HIP events to find the dominant kernel
    engine.set_profiling(vello_amd.renderer.STAGES)
    for _ in range(max(args.warmup, 1)):
        pipe.step()
    pipe.flush()
    torch.cuda.synchronize()
    rc = engine.sync()
    if rc != 0:
        raise SystemExit(f"frame failed: {rc} {engine.bump()}")
    engine.stage_ms()
    engine.kernel_ms()  # (both read-and-reset: what follows is measured one frame at a time)
    bump = engine.bump()
    slice_items, cov_words = engine.fine_slice_stats()  # fine's long tiles cut into slices (engine.h FINE_SLICE_FILLS)
    # the dominant kernel = the stage with the longest launch when frames run one at a time (with frames in flight a
    # stage's events also span the other frames' kernels that share the CUs with it)
    for _ in range(10):
        engine.render_resident(WIDTH, HEIGHT, BASE_COLOR, aa, out=frame)
        engine.sync_frame(0)
    engine.sync()
    iso_ms = engine.stage_ms()
    # ... kernel by kernel: flatten and coarse are stages of several kernels (vello_hip_get_kernel_ms times them apart)
    iso_k = engine.kernel_ms()
    dominant_kernel, dominant, _, _ = dominant_of(engine, iso_ms, iso_k)  # (its stage: what the timed region's events go around)

    # timed region: exactly K steps, events only around the dominant kernel (+ one completion event per frame)
    engine.set_profiling([dominant])
    step_done.clear()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.step()
    pipe.flush()
    rc = engine.sync()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if rc != 0:
        raise SystemExit(f"frame failed in the timed region: {rc} {engine.bump()}")
    dom_ms, dom_n = engine.stage_ms()[dominant]
    if dominant in engine.KERNELS:  # (the kernel's own share of its stage)
        dom_ms, dom_n = engine.kernel_ms()[dominant_kernel]
    # completion-to-completion intervals of consecutive frames (each step waits for the oldest frame in flight)
    intervals = [(b - a) * 1e3 for a, b in zip(step_done[:-1], step_done[1:])]
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    own_fps = steps / elapsed

    # the same pass once more over a FIXED 200 steps (">= 200 frames"): the driver's --steps is its own
    fps_200 = None
    if timed_headline and not distributed and not args.timed_only:
        engine.set_profiling([])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(200):
            pipe.step()
        pipe.flush()
        engine.sync()
        fps_200 = 200 / (time.perf_counter() - t1)
        step_done_200 = list(step_done[-200:])
        intervals_200 = [(b - a) * 1e3 for a, b in zip(step_done_200[:-1], step_done_200[1:])]
    else:
        intervals_200 = []

    exchange_ms = None
    if distributed and timed_headline:
        # the exchange step on its own ("report the gather time separately"): nothing else on the GPUs
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(10):
            gather_frames(ring[0], rank, world, dst=0, out=gathered)
        torch.cuda.synchronize()
        dist.barrier()
        exchange_ms = (time.perf_counter() - t1) / 10 * 1e3

    # one frame at a time: frame latency (host clock around render + wait) and the isolated per-kernel durations
    n_serial = 0 if args.timed_only else (200 if timed_headline else 60)  # (d1: the median of >= 200 frames)

    def profiled_frames(n):
        """n frames one at a time with the enabled stages under HIP events, read in five batches: per stage and kernel the MEDIAN
        of the batches' mean launch times (the engine reports sums; one frame that meets a stall -- a first-use allocation, a
        clock step -- would otherwise sit in the mean of a short run), as (ms, 1) pairs."""
        per = max(n // 5, 2) if n else 0
        got_s, got_k = [], []
        for _ in range(5 if n else 0):
            for _ in range(per):
                engine.render_resident(WIDTH, HEIGHT, BASE_COLOR, aa, out=frame)
                engine.sync_frame(0)
            engine.sync()
            got_s.append(engine.stage_ms())
            got_k.append(engine.kernel_ms())
        med = lambda rows, k: (sorted(r[k][0] / max(r[k][1], 1) for r in rows)[len(rows) // 2], 1)
        if not got_s:
            return engine.stage_ms(), engine.kernel_ms()
        return {k: med(got_s, k) for k in got_s[0]}, {k: med(got_k, k) for k in got_k[0]}

    engine.set_profiling([])
    serial = []
    lat_ms, lat_k = {}, {}
    if n_serial:
        # ... as a client that wants latency configures the engine: ONE frame in flight (flatten then runs its stroke
        # workgroups beside the heavy list's instead of before them, engine.h Frame::flatten_side_by_side)
        engine.set_frames_in_flight(1)
        for _ in range(n_serial):
            t1 = time.perf_counter()
            engine.render_resident(WIDTH, HEIGHT, BASE_COLOR, aa, out=frame)
            engine.sync_frame(0)
            serial.append((time.perf_counter() - t1) * 1e3)
        engine.set_profiling(vello_amd.renderer.STAGES)
        lat_ms, lat_k = profiled_frames(min(n_serial, 30))
        engine.set_profiling([])
        engine.set_frames_in_flight(nif)
    # the kernels of the timed region's configuration, one frame at a time
    engine.set_profiling(vello_amd.renderer.STAGES)
    all_ms, all_k = profiled_frames(min(n_serial, 50))
    engine.set_profiling([])
    if args.timed_only:
        all_ms = {k: (0.0, 0) for k in vello_amd.renderer.STAGES}
        all_ms[dominant] = (dom_ms, dom_n)
        all_k = {k: (0.0, 0) for k in all_k}
        if dominant in engine.KERNELS:
            all_k[dominant_kernel] = (dom_ms, dom_n)

    # PCIe-inclusive rates (never `value`): the packed scene starts in host memory every frame.
    pcie_fps = pcie_pipelined_fps = None
    if timed_headline and not args.timed_only:
        n_pcie = 20
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_pcie):  # (a) serial: upload, render, wait -- what a blocking render_to_texture does
            engine.upload_scene(wl.packed, wl.layout)
            engine.render_resident(WIDTH, HEIGHT, BASE_COLOR, aa, out=frame)
            engine.sync()
        pcie_fps = n_pcie / max(time.perf_counter() - t1, 1e-9)
        n_pipe = 60  # (b) vello_hip_render_frame: the H2D copy of frame i+1 overlaps the kernels of the frames before it
        for i in range(min(n_pipe, nif)):
            engine.render_frame(wl.packed, wl.layout, WIDTH, HEIGHT, BASE_COLOR, aa, out=ring[i % nif])
        engine.sync()
        t1 = time.perf_counter()
        for i in range(n_pipe):
            engine.render_frame(wl.packed, wl.layout, WIDTH, HEIGHT, BASE_COLOR, aa, out=ring[i % nif])
        engine.sync()
        pcie_pipelined_fps = n_pipe / max(time.perf_counter() - t1, 1e-9)
        engine.upload_scene(wl.packed, wl.layout)
        engine.render_resident(WIDTH, HEIGHT, BASE_COLOR, aa, out=frame)
        engine.sync()

    ptcl_words = 64 * ((WIDTH + 15) // 16) * ((HEIGHT + 15) // 16) + bump["ptcl"]
    sb = stage_bytes(wl.layout, bump, wl.n_tag_words, WIDTH, HEIGHT, ptcl_words)
    frame_bytes = d4_frame_bytes(wl.packed.nbytes, wl.layout, bump, wl.n_tag_words, WIDTH, HEIGHT, ptcl_words)
    stage_iso_ms = all_ms[dominant][0] / max(all_ms[dominant][1], 1)
    if dominant in engine.KERNELS:
        iso_ms = all_k[dominant_kernel][0] / max(all_k[dominant_kernel][1], 1)
    else:
        iso_ms = stage_iso_ms
    ovl_ms = dom_ms / max(dom_n, 1)
    # Round 6: several kernels have a form for ONE frame in flight and a form for several (k_fine: 4 waves per SIMD and long tiles
    # sliced / 5 waves and unsliced; k_path_count: 48.7 / 24.6 KB of LDS) -- the in-flight forms are built to share the chip and are
    # slower when they have it to themselves.  `frac` is the dominant kernel with the chip to itself in the form built for that
    # (one frame in flight: what rocprofv3 shows for `bench.py --in-flight 1 --timed-only`, profiles/r0N_kernel_stats_serial_d2.csv);
    # the in-flight form's isolated and overlapped durations are reported beside it.
    form_iso_ms, form_stage_iso_ms = iso_ms, stage_iso_ms  # (the timed region's configuration, launched one at a time)
    if lat_ms and dominant in lat_ms and lat_ms[dominant][1]:
        stage_iso_ms = lat_ms[dominant][0] / max(lat_ms[dominant][1], 1)
        iso_ms = (lat_k[dominant_kernel][0] / max(lat_k[dominant_kernel][1], 1)) if dominant in engine.KERNELS else stage_iso_ms
    form_frac_ms = form_stage_iso_ms if dominant in engine.KERNELS else form_iso_ms
    #  algorithmic bytes exist per STAGE: when the dominant kernel is one of a stage's several, `achieved` / `frac`
    # are the stage's bytes over the stage's time (all its kernels), not over the one kernel's
    frac_ms = stage_iso_ms if dominant in engine.KERNELS else iso_ms
    tile_area_bytes = 36 * bump["tile"]
    res = {
        "engine": engine, "frame": frame, "steps": steps, "elapsed": elapsed, "own_fps": own_fps, "bump": bump, "dominant": dominant, "dominant_kernel": dominant_kernel,
        "fine_slices": {"slice_work_items": slice_items, "coverage_scratch_bytes": cov_words * 4,
                        "rule": "MSAA: a tile of >= 96 FILLs (one frame in flight; >= 192 with frames in flight: the timed region slices nothing on d2) is cut into slices of 32 fills (coverage by one wave per slice, composited by the last to finish)"},
        "exchange_ms": exchange_ms, "pcie_fps": pcie_fps, "pcie_pipelined_fps": pcie_pipelined_fps,
        "describe": wl.describe(engine),
        "frame_ms": {"median": pct(intervals, 0.5), "p10": pct(intervals, 0.1), "p90": pct(intervals, 0.9), "mean": (sum(intervals) / len(intervals)) if intervals else None,
                     "n": len(intervals)},
        "fps_200": fps_200,
        "frame_ms_200": {"median": pct(intervals_200, 0.5), "p10": pct(intervals_200, 0.1), "p90": pct(intervals_200, 0.9),
                         "mean": (sum(intervals_200) / len(intervals_200)) if intervals_200 else None, "n": len(intervals_200)},
        "serial_ms": {"median": pct(serial, 0.5), "p10": pct(serial, 0.1), "p90": pct(serial, 0.9), "n": len(serial)},
        "roofline": {
            "bound": "hbm",
            "kernel": dominant_kernel,
            "kernel_how": "the kernel with the longest isolated launch (HIP events around every kernel, one frame at a time; flatten and coarse "
                          "are stages of three / two kernels, timed apart)",
            "achieved_and_frac_of": ("the kernel's STAGE: its algorithmic bytes over the isolated time of all its kernels (bytes exist per stage only)"
                                     if dominant in engine.KERNELS else "the kernel: its algorithmic bytes over its isolated launch time"),
            "achieved": round(sb[dominant] / (frac_ms * 1e-3) / 1e9, 2) if frac_ms > 0 else None,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(sb[dominant] / (frac_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if frac_ms > 0 else None,
            "algorithmic_bytes_per_launch": int(sb[dominant]),
            "avg_launch_ms": round(iso_ms, 5),
            "avg_launch_ms_how": ("HIP events around the kernel on the engine's stream, one frame at a time with ONE frame in flight configured (the form built for "
                                  "having the chip to itself; rocprofv3 of `bench.py --in-flight 1 --timed-only` shows the same kernel): median of five batches' mean "
                                  "launch times" if lat_ms else "HIP events around the kernel in the timed region"),
            "in_flight_form": {
                "what": "the form of the dominant kernel the timed region launches (frames in flight: k_fine for five waves per SIMD, long tiles not sliced), "
                        "launched one frame at a time, and under the overlap of the timed region",
                "avg_launch_ms_alone": round(form_iso_ms, 5),
                "frac_alone": round(sb[dominant] / (form_frac_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if form_frac_ms > 0 else None,
            },
            "avg_launch_ms_overlapped": round(ovl_ms, 5),
            "achieved_overlapped": round(sb[dominant] / (ovl_ms * 1e-3) / 1e9, 2) if ovl_ms > 0 else None,
            "frac_overlapped": round(sb[dominant] / (ovl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ovl_ms > 0 else None,
            "frame_algorithmic_bytes": int(frame_bytes),
            "frame_algorithmic_formula": "SURVEY 8d d4: S + 48Tw + 96P + 168D + 8Bd + 48L + 36A + 48C + 48G + 8Wp + 4Npx",
            "frame_achieved_GBps": round(frame_bytes / (elapsed / steps) / 1e9, 2),
            "frame_frac": round(frame_bytes / (elapsed / steps) / 1e9 / HBM_PEAK_GBS, 5),
            "frame_frac_without_tile_area": round((frame_bytes - tile_area_bytes) / (elapsed / steps) / 1e9 / HBM_PEAK_GBS, 5),
            "frame_frac_without_tile_area_how": "d4 minus its 36 A term: the passes over every path's bounding-box tiles (zero fill, backdrop, bit planes), whose size has nothing to do with what is drawn",
            "frame_achieved_GBps_one_frame_at_a_time": round(frame_bytes / (pct(serial, 0.5) * 1e-3) / 1e9, 2) if serial else None,
            "stage_ms": {k: round(v[0] / max(v[1], 1), 5) for k, v in all_ms.items()},
            "kernel_ms_of_multi_kernel_stages": {k: round(v[0] / max(v[1], 1), 5) for k, v in all_k.items()},
            "stage_ms_one_frame_in_flight": {k: round(v[0] / max(v[1], 1), 5) for k, v in lat_ms.items()},
            "kernel_ms_one_frame_in_flight": {k: round(v[0] / max(v[1], 1), 5) for k, v in lat_k.items()},
            "stage_algorithmic_bytes": {k: int(v) for k, v in sb.items()},
        },
    }
    return res


def cpu_baseline(wl, frame):
    """The CPU oracle on the same workload, bounded: ~10-25 s of CPU work.  Single thread, then THREADED_STAGES
    threaded on every host core."""
    from oracle.oracle import Oracle

    o = Oracle(capacity_scale=8)
    o.set_scene(wl.packed, wl.layout, WIDTH, HEIGHT, BASE_COLOR, 2)
    t0 = time.perf_counter()
    ref = o.render()  # also the warm-up
    one = time.perf_counter() - t0
    n_cpu = max(2, min(10, int(10.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n_cpu):
        ref = o.render()
    cpu_s = (time.perf_counter() - t0) / n_cpu
    same = bool(np.array_equal(ref, frame.cpu().numpy()))
    n_thr = max(1, os.cpu_count() or 1)
    o.set_threads(n_thr)
    o.render()
    t0 = time.perf_counter()
    n_mt = 0
    while n_mt < 3 or (time.perf_counter() - t0 < 8.0 and n_mt < 40):  # bounded sample: ~8 s, at least 3 frames
        ref_mt = o.render()
        n_mt += 1
    cpu_mt_s = (time.perf_counter() - t0) / n_mt
    same_mt = bool(np.array_equal(ref_mt, ref))
    cargo = None
    try:
        cargo = subprocess.check_output(["cargo", "--version"], stderr=subprocess.DEVNULL, timeout=10).decode().strip()
    except Exception:
        pass
    return {
        "value": round(1.0 / cpu_mt_s, 4),
        "unit": "frames/s",
        "cores": n_thr,
        "cores_note": "fine on all of them; flatten / path_count / coarse / path_tiling on at most 32 (shared atomics stop scaling)",
        "value_single_thread": round(1.0 / cpu_s, 4),
        "kind": "port",
        "sample": f"{n_mt} full frames of the '{wl.key}' scene on {n_thr} threads (+ {n_cpu} on one thread): C restatement of "
                  f"vello_shaders/src/cpu + fine.wgsl (NOT vello_cpu; cargo on this box: {cargo or 'absent'}); threaded stages: "
                  f"{THREADED_STAGES}; output identical to the GPU frame: {same}; threaded == single-thread output: {same_mt}",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--in-flight", type=int, default=4,
                    help="frames the engine keeps in flight (wgpu queues recordings the same way); 1 = serial frames")
    ap.add_argument("--workload", choices=["both", "d2", "r1mix"], default="both",
                    help="d2 = SURVEY 8d d2's C3 scene (the workload of `value`); r1mix = round 1's mix; both = d2 + r1mix beside it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip config.other_configs (BASELINE's C2 / C4, ~1 s each)")
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

    head_key = "r1mix" if args.workload == "r1mix" else "d2"
    head_wl = Workload(head_key, rank)
    peak_measured = measure_copy_peak(local_rank) if rank == 0 else None
    head = run_workload(head_wl, args, rank, local_rank, world, timed_headline=True)
    second = None
    if args.workload == "both" and not distributed and not args.timed_only:
        head["engine"] = None  # free the first context's pools before the second one allocates
        frame_head = head["frame"]
        second_wl = Workload("r1mix", rank)
        second = run_workload(second_wl, args, rank, local_rank, world, timed_headline=False)
        second["engine"] = None
    else:
        frame_head = head["frame"]

    # per-rank frames/s (each rank's own clock over the same K steps), for the scaling record
    per_rank = None
    if distributed:
        t = torch.zeros(world, device=f"cuda:{local_rank}", dtype=torch.float64)
        t[rank] = head["own_fps"]
        dist.all_reduce(t)
        per_rank = [round(float(x), 2) for x in t.tolist()]

    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    steps, elapsed = head["steps"], head["elapsed"]
    ms_per_step = elapsed / steps * 1e3
    value = world * steps / elapsed
    dominant = head["dominant"]

    # HBM traffic of the dominant kernel from the PMC passes (collected separately, as the pool requires, by
    # scripts/gpu_calib.sh; corrected with the factors calibrated in the same session; stamped with its commit)
    traffic = traffic_commit = traffic_sources = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            tj = json.load(fh)
        traffic = tj.get("workloads", {}).get(head_key, tj).get("kernels", {}).get(head["dominant_kernel"], {}).get("traffic_bytes")
        traffic_commit = tj.get("commit")
        traffic_sources = tj.get("kernel_sources")
        if second is not None:  # (the other workload's dominant kernel, same passes)
            second["roofline"]["traffic"] = (tj.get("workloads", {}).get("r1mix", {}).get("kernels", {})
                                             .get(second["dominant_kernel"], {}).get("traffic_bytes"))
    except (OSError, ValueError):
        pass

    roof = head["roofline"]
    roof["peak_measured"] = round(peak_measured, 1) if peak_measured else None
    roof["peak_measured_how"] = "float4 device-to-device copy kernel (scripts/calib/copy_bw.hip) of 1 GiB on this GPU in this run, read + written bytes / best of 5"
    roof["traffic"] = traffic
    # (the PMC passes are a session of their own; say so when the kernels have changed since.  Compared by a hash
    # of the kernel sources -- a commit hash moves with every documentation commit and the GPU box has no .git)
    from vello_amd._lib import kernel_sources_hash

    sources_now = kernel_sources_hash()
    stale = traffic is not None and traffic_sources != sources_now
    roof["traffic_stale"] = bool(stale)
    roof["traffic_source"] = ("profiles/pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per launch, "
                              f"collected at commit {traffic_commit}, kernel sources {traffic_sources}; this run's kernel sources {sources_now}: "
                              + ("STALE -- the kernels have changed since the PMC passes)" if stale else "the same kernels)"))
    serial_med = head["serial_ms"]["median"]
    result = {
        "metric": "frames/sec paris-30k 1600x1600 MSAA16; scenes/sec at 1/2/4/8 GPU",
        "value": round(value, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32+u32",
        "data": "synthetic",
        "config": {
            "workload": head["describe"],
            "baseline_config": "configs[2]",
            "commit": git_head(),
            "fine_kernel": "k_fine",
            "fine_slices": head["fine_slices"],
            "parallelism": f"scenes{world}" if distributed else "single",
            "exchange": "RCCL gather of RGBA8 frames to rank 0 each step" if distributed else "none",
            "exchange_alone_ms": None if head["exchange_ms"] is None else round(head["exchange_ms"], 4),
            "per_rank_frames_per_s": per_rank,
            "bump": head["bump"],
            "frames_in_flight": max(1, min(args.in_flight, 8)),
            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "frame_completion_interval_ms_pipelined": head["frame_ms"],
            "frame_completion_interval_how": "host clock between consecutive frames' completions with --in-flight frames queued; completions come in bursts "
                                             "(the frames in flight finish close together), so the MEDIAN interval is well below ms_per_step -- the mean is ms_per_step",
            "value_200_steps": None if head["fps_200"] is None else round(head["fps_200"], 2),
            "frame_completion_interval_ms_200_steps": head["frame_ms_200"],
            "frame_ms_one_at_a_time": head["serial_ms"],
            "value_one_frame_at_a_time": round(1e3 / serial_med, 2) if serial_med else None,
            "one_frame_at_a_time_how": "vello_hip_set_frames_in_flight(1), then render + wait per frame on the host clock, median of 200 frames (SURVEY 8d d1 / b5: "
                                       "one frame in flight per renderer); NOTE a different kernel configuration than `value`'s: with one frame in flight flatten "
                                       "runs k_flatten_main / k_flatten_tail instead of k_flatten_strokes / k_flatten_heavy",
            "pcie_inclusive_frames_per_s": None if head["pcie_fps"] is None else round(head["pcie_fps"], 2),
            "pcie_inclusive_pipelined_frames_per_s": None if head["pcie_pipelined_fps"] is None else round(head["pcie_pipelined_fps"], 2),
        },
        "roofline": roof,
    }
    if second is not None:
        smed = second["serial_ms"]["median"]
        result["config"]["secondary"] = {
            "workload": second["describe"],
            "value": round(second["steps"] / second["elapsed"], 2),
            "steps": second["steps"],
            "ms_per_step": round(second["elapsed"] / second["steps"] * 1e3, 4),
            "value_one_frame_at_a_time": round(1e3 / smed, 2) if smed else None,
            "frame_completion_interval_ms_pipelined": second["frame_ms"],
            "frame_ms_one_at_a_time": second["serial_ms"],
            "bump": second["bump"],
            "fine_slices": second["fine_slices"],
            "roofline": second["roofline"],
        }
    if world == 1 and not args.timed_only and not args.no_other_configs:
        others = []
        for key, label in (("tiger", "configs[1]: Ghostscript_Tiger.svg, 1024x1024, MSAA8"), ("mmark", "configs[3]: mmark-style 50k stroked quads, 2048x2048, MSAA16")):
            try:
                others.append(measure_other_config(key, label, local_rank))
            except Exception as e:  # (never lose the headline line to a side measurement)
                others.append({"config": label, "error": repr(e)})
        result["config"]["other_configs"] = others
    if world == 1 and not args.no_cpu_baseline and not args.timed_only:
        result["cpu_baseline"] = cpu_baseline(head_wl, frame_head)
    print(json.dumps(result))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
