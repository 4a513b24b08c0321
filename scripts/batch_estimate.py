"""What would stage-synchronous batching of N frames buy?  One launch per stage over the work of N frames
is, to the hardware, one frame of N x the work: d2's generator at 4 x the paths on 4 x the area (the same paths per tile, the
same path sizes) rendered ONE frame at a time is that launch shape without touching the engine.  Compared: 4 x (d2 alone),
d2 with four frames in flight (today's `value`), and the 4 x scene alone -- per stage.
   python scripts/batch_estimate.py"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vello_amd, workloads, bench
from vello_amd import AaConfig
from vello_amd.renderer import STAGES


def run(name, scene, size, caps, nif_list=(1, 4)):
    packed, layout = scene.resolve()
    eng = vello_amd.Engine(0, 4, caps)
    eng.upload_scene(packed, layout)
    out = [torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda:0") for _ in range(4)]
    torch.cuda.synchronize()
    res = {"workload": name, "size": size, "paths": layout.n_paths}
    for nif in nif_list:
        eng.set_frames_in_flight(nif)
        for i in range(8):
            eng.render_resident(size, size, bench.BASE_COLOR, AaConfig.Msaa16, out=out[i % 4])
        assert eng.sync() == 0, eng.bump()
        n = 100 if size <= 1600 else 40
        t = time.perf_counter()
        for i in range(n):
            eng.render_resident(size, size, bench.BASE_COLOR, AaConfig.Msaa16, out=out[i % 4])
            if nif == 1:
                eng.sync_frame(0)
        assert eng.sync() == 0
        res[f"ms_per_frame_{nif}_in_flight"] = round((time.perf_counter() - t) / n * 1e3, 4)
    eng.set_frames_in_flight(1)
    eng.set_profiling(STAGES)
    eng.stage_ms(); eng.kernel_ms()
    for i in range(10):
        eng.render_resident(size, size, bench.BASE_COLOR, AaConfig.Msaa16, out=out[0]); eng.sync_frame(0)
    ms, km = eng.stage_ms(), eng.kernel_ms()
    res["stage_us"] = {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in ms.items()}
    res["kernel_us"] = {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in km.items()}
    res["sum_us"] = round(sum(res["stage_us"].values()), 1)
    res["bump"] = eng.bump()
    print(json.dumps(res), flush=True)
    return res


if __name__ == "__main__":
    a = run("d2", workloads.paris_like_scene_d2(bench.SEED0), 1600, dict(bench.D2_CAPS))
    caps4 = {k: min(v * 4, (1 << 27)) for k, v in bench.D2_CAPS.items()}
    caps4["bin_data"] = 1 << 22  # (info + bin entries: 519 436 bin entries for the 4 x scene, the default pool is the reference's 2^18)
    b = run("d2 x 4 (120 000 paths, 3200^2)", workloads.paris_like_scene_d2(bench.SEED0, n_paths=120000, size=3200.0), 3200, caps4, nif_list=(1,))
    print("per stage, us: d2 alone x 4  ->  the 4 x scene alone   (ratio)")
    for k in a["stage_us"]:
        print(f"  {k:14s} {4 * a['stage_us'][k]:8.1f} -> {b['stage_us'][k]:8.1f}   ({b['stage_us'][k] / max(4 * a['stage_us'][k], 1e-9):.2f})")
    print(f"  one d2 frame: alone {a['ms_per_frame_1_in_flight']} ms, four in flight {a['ms_per_frame_4_in_flight']} ms; "
          f"a quarter of the 4 x scene alone: {b['ms_per_frame_1_in_flight'] / 4:.4f} ms")
