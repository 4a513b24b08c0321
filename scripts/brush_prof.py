"""The brush / blend-stack specialisation of fine, k_fine<2, true> (f1 / f3 are correct and were never
profiled): blend_grid, gradient_extend and image_sampling at stated sizes, MSAA16.
   python scripts/brush_prof.py stages     -> per-stage / per-kernel times of the PRODUCT library, one frame at a time, + frames/s
   python scripts/brush_prof.py phases     -> k_fine's phase table (measurement build ab_tmp/libvello_hip_PROF.so, scripts/build_prof.sh)
   (one mode per process: the library is chosen at import)"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "stages"
if mode == "phases":
    import fine_prof  # (selects the PROF library)
elif os.environ.get("VELLO_AB_LIB"):
    import vello_amd._lib as _L
    _L._use_library(os.path.join(ROOT, "ab_tmp", "libvello_hip_%s.so" % os.environ["VELLO_AB_LIB"]))
import numpy as np, torch
import vello_amd, workloads
from vello_amd import AaConfig
from vello_amd.renderer import STAGES


def scenes():
    out = []
    s, w, h = workloads.blend_grid_scene()
    out.append(("blend_grid 900x900", s, w, h))
    for name, fn in (("gradient_extend", workloads.gradient_extend_scene), ("image_sampling", workloads.image_sampling_scene)):
        r = fn()
        s, w, h = r if isinstance(r, tuple) else (r, 512, 512)
        out.append((f"{name} {w}x{h}", s, w, h))
    return out


def stages(name, s, w, h):
    r = vello_amd.Resolver().resolve(s)
    eng = vello_amd.Engine()
    eng.upload_resolved(r)
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0"); torch.cuda.synchronize()
    for _ in range(5):
        eng.render_resident(w, h, 0xFFFFFFFF, AaConfig.Msaa16, out=out); eng.sync_frame(0)
    t = time.perf_counter()
    for _ in range(100):
        eng.render_resident(w, h, 0xFFFFFFFF, AaConfig.Msaa16, out=out); eng.sync_frame(0)
    lat = (time.perf_counter() - t) / 100 * 1e3
    eng.set_profiling(STAGES)
    eng.stage_ms(); eng.kernel_ms()
    for _ in range(30):
        eng.render_resident(w, h, 0xFFFFFFFF, AaConfig.Msaa16, out=out); eng.sync_frame(0)
    ms, km = eng.stage_ms(), eng.kernel_ms()
    row = {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in ms.items()}
    print(json.dumps({"workload": name, "aa": "msaa16", "one_frame_ms": round(lat, 4), "stage_us": row, "sum_us": round(sum(row.values()), 1),
                      "kernel_us": {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in km.items()}, "bump": eng.bump()}), flush=True)


def phases(name, s, w, h):
    r = vello_amd.Resolver().resolve(s)
    eng = vello_amd.Engine()
    eng.upload_resolved(r)
    fine_prof.report_engine(name, eng, w, h, 2)


if __name__ == "__main__":
    for name, s, w, h in scenes():
        (phases if mode == "phases" else stages)(name, s, w, h)
