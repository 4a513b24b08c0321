"""Round 5, second sitting: where the zero fill of a frame's tiles rides -- in k_pathtag_scan's launch (the tree's default), in
k_flatten_light's (VELLO_HIP_PREZERO_AT=light), or in tile_alloc as before (VELLO_HIP_DEBUG_NO_PREZERO) -- ONE context per scene, the three
settings alternating on it.     python scripts/round5b_ab3.py [reps]"""
import json
import os
import statistics
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sys.argv = [sys.argv[0], str(reps)]
sys.path.insert(0, os.path.join(ROOT, "scripts", "experiments"))
import round5b_ab as R  # noqa: E402

for key in R.SCENES:
    wl = R.make_workload(key)
    ring = R.make_ring(wl)
    e = R.make_engine("A", wl.caps)
    e.upload_scene(wl.packed, wl.layout)
    rows = {}
    for rep in range(reps):
        for setting in ("scan", "light", "tile_alloc"):
            os.environ["VELLO_HIP_PREZERO_AT"] = "light" if setting == "light" else "scan"
            e.set_debug_flags(no_prezero=(setting == "tile_alloc"))
            r = R.measure(e, wl, ring)
            r.update({"scene": key, "zero_fill_in": setting, "rep": rep})
            print(json.dumps(r), flush=True)
            rows.setdefault(setting, []).append(r)
    for setting, rs in rows.items():
        st = {n: statistics.median([r["stage_us"][n] for r in rs]) for n in ("pathtag_scan", "flatten", "tile_alloc")}
        sys.stderr.write("%-6s zero fill in %-10s %6.0f frames/s %6.1f us | %s | light %.1f | %d tiles\n" % (
            key, setting, statistics.median([f for r in rs for f in r["fps_4_in_flight"]]), statistics.median([r["latency_us"] for r in rs]),
            " ".join("%s %.1f" % (n, x) for n, x in st.items()), statistics.median([r["kernel_us"]["k_flatten_light"] for r in rs]), rs[0]["prezero_tiles"]))
    del e
os.environ.pop("VELLO_HIP_PREZERO_AT", None)
