"""Round 5, second sitting, follow-up to round5b_ab.py: is what separated P from C / A on d2's UNCHANGED kernels (k_flatten_main, k_flatten_tail,
k_path_count: identical ISA in both libraries) the library or the order in which the contexts were created?  d2 only; contexts created in
the order given on the command line (default A P A P), each measured rep by rep; a context of the tree's library is measured both with
the tiles zeroed beside k_flatten_light and with VELLO_HIP_DEBUG_NO_PREZERO (the same buffers, the same code object).

    python scripts/round5b_ab2.py [order] [reps]"""
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + sys.argv[1:]
order = sys.argv[1] if len(sys.argv) > 1 else "APAP"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sys.argv = [sys.argv[0], str(reps)]
sys.path.insert(0, os.path.join(ROOT, "scripts", "experiments"))
import round5b_ab as R  # noqa: E402

wl = R.make_workload("d2")
ring = R.make_ring(wl)
engines = []
for i, v in enumerate(order):
    e = R.make_engine(v, wl.caps)
    e.upload_scene(wl.packed, wl.layout)
    engines.append(("%s%d" % (v, i), v, e))
rows = {}
for rep in range(reps):
    for name, v, e in engines:
        for flag in ((False, True) if v == "A" else (False,)):
            if v == "A":
                e.set_debug_flags(no_prezero=flag)
            r = R.measure(e, wl, ring)
            key = name + ("-noprezero" if flag else "")
            r.update({"context": key, "rep": rep})
            print(json.dumps(r), flush=True)
            rows.setdefault(key, []).append(r)
import statistics  # noqa: E402
for key, rs in rows.items():
    k = {n: statistics.median([r["kernel_us"][n] for r in rs]) for n in rs[0]["kernel_us"]}
    st = {n: statistics.median([r["stage_us"][n] for r in rs]) for n in ("tile_alloc", "path_count", "fine", "path_tiling", "backdrop", "pathtag_scan")}
    sys.stderr.write("%-14s %6.0f frames/s %6.1f us | %s | %s\n" % (key, statistics.median([f for r in rs for f in r["fps_4_in_flight"]]),
                     statistics.median([r["latency_us"] for r in rs]), " ".join("%s %.1f" % (n[2:], x) for n, x in k.items()),
                     " ".join("%s %.1f" % (n, x) for n, x in st.items())))
