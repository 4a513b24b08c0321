"""Small scenes: what sharing launches buys (k_front, flatten.hip).  Same box, same process, interleaved: every
workload with the stages of a small scene fused into launches (the default) and with every stage as a kernel of its own
(VELLO_HIP_DEBUG_NO_FUSION) -- one frame at a time (enqueue + wait, ms) and four frames in flight (frames/s).
   python scripts/small_scene_latency.py [rounds]  -> one JSON line per workload"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if os.environ.get("VELLO_AB_LIB"):
    import vello_amd._lib as _L
    _L._use_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ab_tmp", "libvello_hip_%s.so" % os.environ["VELLO_AB_LIB"]))
import vello_amd, workloads
from vello_amd import AaConfig

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def measure(eng, w, h, aa, ring):
    for i in range(30):
        eng.render_resident(w, h, 0xFFFFFFFF, aa, out=ring[0]); eng.sync_frame(0)
    t = time.perf_counter()
    for i in range(300):
        eng.render_resident(w, h, 0xFFFFFFFF, aa, out=ring[0]); eng.sync_frame(0)
    return (time.perf_counter() - t) / 300 * 1e3


def run(name, packed, layout, w, h, aa, resolved=None):
    out = {"workload": name}
    for nif in (1, 4):
        eng = vello_amd.Engine()
        eng.set_frames_in_flight(nif)
        if resolved is not None:
            eng.upload_resolved(resolved)
        else:
            eng.upload_scene(packed, layout)
        ring = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0") for _ in range(nif)]
        torch.cuda.synchronize()
        res = {"fused": [], "unfused": []}
        for r in range(ROUNDS):
            for which in ("fused", "unfused"):
                eng.set_debug_flags(no_fusion=which == "unfused")
                if nif == 1:
                    res[which].append(measure(eng, w, h, aa, ring))
                else:
                    for i in range(40):
                        eng.render_resident(w, h, 0xFFFFFFFF, aa, out=ring[i % nif])
                    assert eng.sync() == 0
                    t = time.perf_counter()
                    for i in range(400):
                        eng.render_resident(w, h, 0xFFFFFFFF, aa, out=ring[i % nif])
                    assert eng.sync() == 0
                    res[which].append(400 / (time.perf_counter() - t))
        key = "one_frame_ms" if nif == 1 else "frames_per_s_4_in_flight"
        out[key] = {k: round(float(np.median(v)), 4 if nif == 1 else 0) for k, v in res.items()}
        out[key + "_all"] = {k: [round(float(x), 4 if nif == 1 else 0) for x in v] for k, v in res.items()}
        if nif == 1:
            out["fused_launches_per_frame"] = None
            eng.set_debug_flags()
            b = eng.fused_launches()
            eng.render_resident(w, h, 0xFFFFFFFF, aa, out=ring[0]); eng.sync_frame(0)
            out["fused_launches_per_frame"] = eng.fused_launches() - b
    print(json.dumps(out), flush=True)


d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tiger_scene.npz"))
run("C1 circle 256^2 area", *workloads.circle_scene().resolve(), 256, 256, AaConfig.Area)
run("C2 tiger 1024^2 MSAA8", d["packed"], vello_amd.Layout(*[int(v) for v in d["layout"]]), 1024, 1024, AaConfig.Msaa8)
run("stroke_styles 256^2 MSAA16", *workloads.stroke_styles_scene().resolve(), 256, 256, AaConfig.Msaa16)
s, w, h = workloads.blend_grid_scene(); r = vello_amd.Resolver().resolve(s)
run("blend_grid 900^2 MSAA16", r.packed, r.layout, w, h, AaConfig.Msaa16, resolved=r)
s, w, h = workloads.gradient_extend_scene(); r = vello_amd.Resolver().resolve(s)
run("gradient_extend %dx%d MSAA16" % (w, h), r.packed, r.layout, w, h, AaConfig.Msaa16, resolved=r)
s, w, h = workloads.image_sampling_scene(); r = vello_amd.Resolver().resolve(s)
run("image_sampling %dx%d MSAA16" % (w, h), r.packed, r.layout, w, h, AaConfig.Msaa16, resolved=r)
p, l = workloads.random_test_scene(5, n_paths=2000, size=1024.0, strokes=True, clips=True).resolve()
run("random 2000 paths 1024^2 MSAA16", p, l, 1024, 1024, AaConfig.Msaa16)
