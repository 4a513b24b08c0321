"""Per-phase cycle counts of k_coarse (measurement build: make -C vello_amd/csrc EXTRA=-DVELLO_COARSE_PROF).
Renders one frame of a bench workload and prints what each workgroup's wave 0 spent where."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vello_amd._lib as L
L._use_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ab_tmp", "libvello_hip_PROF.so"))
import bench
from vello_amd.renderer import Engine

for key in sys.argv[1:] or ["d2", "r1mix"]:
    wl = bench.Workload(key, 0)
    eng = Engine(0, 4, wl.caps)
    eng.upload_scene(wl.packed, wl.layout)
    for _ in range(3):
        eng.render_resident(bench.WIDTH, bench.HEIGHT, bench.BASE_COLOR, 2)
        eng.sync()
    cap = eng.capacities()["ptcl"]
    whole = eng.read_buffer("ptcl", np.uint32)
    print("ptcl capacity", cap, "buffer words", whole.size)
    raw = whole[cap - 8192:cap].reshape(512, 16)
    act = raw[raw[:, :6].sum(axis=1) > 0]
    tot = act[:, :6].sum(axis=1) + act[:, 8:11].sum(axis=1)
    names = ["stream", "batch front", "allocate", "emit", "batch end", "list end"]
    print(f"{key}: {len(act)} workgroups; cycles per workgroup: mean {tot.mean():.0f}, max {tot.max()} (x 1/2.4 GHz = {tot.max()/2400:.0f} us)")
    worst = act[np.argsort(tot)[-8:]]
    for i, n in enumerate(names):
        print(f"  {n:12s} mean {act[:, i].mean():9.0f}  ({100 * act[:, i].sum() / tot.sum():4.1f} %)   in the 8 slowest workgroups {worst[:, i].mean():9.0f}")
    for i, n in ((8, "  round: issue + fetch_index"), (9, "  round: windows + masks"), (10, "  round: barrier 1"), (0, "  round: queue + barrier 2")):
        print(f"  {n:30s} per round {act[:, i].sum() / max(1, act[:, 6].sum()):8.0f}")
    print(f"  stream rounds mean {act[:, 6].mean():.1f} max {act[:, 6].max()};  batches mean {act[:, 7].mean():.1f} max {act[:, 7].max()}")
    del eng
