"""Round 5, second sitting: the zero fill of a frame's tiles in k_pathtag_scan's launch -- how the stores are made (VELLO_HIP_PREZERO_MODE: 0 plain,
1 plain + a release fence per workgroup, 2 agent-scope 8-byte stores, 3 nontemporal) against tile_alloc doing it (VELLO_HIP_DEBUG_NO_PREZERO).
One context on d2, the settings alternating.     python scripts/round5b_ab4.py [reps]"""
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

wl = R.make_workload("d2")
ring = R.make_ring(wl)
e = R.make_engine("A", wl.caps)
e.upload_scene(wl.packed, wl.layout)
rows = {}
for rep in range(reps):
    for setting in ("tile_alloc", "0", "1", "2", "3"):
        os.environ["VELLO_HIP_PREZERO_MODE"] = setting if setting != "tile_alloc" else "0"
        e.set_debug_flags(no_prezero=(setting == "tile_alloc"))
        r = R.measure(e, wl, ring)
        r.update({"scene": "d2", "mode": setting, "rep": rep})
        print(json.dumps(r), flush=True)
        rows.setdefault(setting, []).append(r)
for setting, rs in rows.items():
    st = {n: statistics.median([r["stage_us"][n] for r in rs]) for n in ("pathtag_scan", "flatten", "tile_alloc", "path_count")}
    sys.stderr.write("d2 zero fill %-10s %6.0f frames/s %6.1f us | %s\n" % (
        setting, statistics.median([f for r in rs for f in r["fps_4_in_flight"]]), statistics.median([r["latency_us"] for r in rs]),
        " ".join("%s %.1f" % (n, x) for n, x in st.items())))
os.environ.pop("VELLO_HIP_PREZERO_MODE", None)
