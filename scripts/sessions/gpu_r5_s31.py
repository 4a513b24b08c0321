"""One process = one setting of the HIP / HSA runtime (environment, before the runtime loads): d2, tiger and the circle, four frames in flight and
one frame at a time (scripts/ab_contexts.py's measurement).   python scripts/sessions/gpu_r5_s31.py LABEL"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
label = sys.argv[1]
sys.argv = ["x"]
import torch
import ab_contexts as R
import bench, vello_amd, workloads
class W: pass
def circle():
    w = W(); w.key = "circle"; w.caps = None; w.width = w.height = 256; w.aa = vello_amd.AaConfig.Area
    w.packed, w.layout = workloads.circle_scene().resolve(); return w
for wl in (bench.Workload("d2", 0), bench.Workload("tiger", 0), circle()):
    ring = [torch.zeros((wl.height, wl.width, 4), dtype=torch.uint8, device="cuda:0") for _ in range(4)]
    torch.cuda.synchronize()
    e = R.make_engine("A", wl.caps); e.upload_scene(wl.packed, wl.layout)
    rs = [R.measure(e, wl, ring) for _ in range(2)]
    print(label, wl.key, "four in flight", [f for r in rs for f in r["fps_4_in_flight"]], "one at a time us", [r["latency_us"] for r in rs], flush=True)
    del e
