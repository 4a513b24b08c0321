"""Per-stage GPU time (HIP events, one frame at a time) on tiger / circle / mmark / paris.
   python scripts/stage_times.py [B]      (B: use ab_tmp/libvello_hip_B.so instead of the in-tree build, see ab_bench.py)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vello_amd, workloads
if len(sys.argv) > 1 and sys.argv[1] != "A":
    import vello_amd._lib as L
    L._use_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ab_tmp", f"libvello_hip_{sys.argv[1]}.so"))
from vello_amd import AaConfig
def run(name, packed, layout, w, h, aa):
    eng = vello_amd.Engine(); eng.upload_scene(packed, layout)
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0"); torch.cuda.synchronize()
    for _ in range(5): eng.render_resident(w, h, 0xFFFFFFFF, aa, out=out); eng.sync_frame(0)
    eng.set_profiling(vello_amd.renderer.STAGES)
    for _ in range(30): eng.render_resident(w, h, 0xFFFFFFFF, aa, out=out); eng.sync_frame(0)
    ms = eng.stage_ms()
    print(name, {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in ms.items()}, "sum_us", round(sum(1e3 * v[0] / max(v[1], 1) for v in ms.values()), 1))
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tiger_scene.npz"))
run("tiger", d["packed"], vello_amd.Layout(*[int(v) for v in d["layout"]]), 1024, 1024, AaConfig.Msaa8)
p, l = workloads.circle_scene().resolve(); run("circle", p, l, 256, 256, AaConfig.Area)
p, l = workloads.mmark_scene().resolve(); run("mmark", p, l, 2048, 2048, AaConfig.Msaa16)
p, l = workloads.paris_like_scene().resolve(); run("paris", p, l, 1600, 1600, AaConfig.Msaa16)
