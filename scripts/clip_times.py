"""GPU time of the clip stage alone (HIP events around its launches): the partitioned kernels of clip.hip against the one-wave
stack machine (VELLO_HIP_DEBUG_SEQ_CLIP), on scenes of nothing but clip layers.   python scripts/clip_times.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vello_amd
from vello_amd import AaConfig
from tests.parity import clip_ops_scene

eng = vello_amd.Engine()
cases = [("600 layers one after another (many_clips)", [1, -1] * 600),
         ("10 000 layers, 4 deep", [1, 1, 1, 1, -1, -1, -1, -1] * 2500),
         ("50 000 layers inside 1 000 open ones", [1] * 1000 + [1, -1] * 50000 + [-1] * 1000),
         ("20 000 layers nested", [1] * 20000 + [-1] * 20000),
         ("250 000 layers, random", list(np.where(np.random.default_rng(5).random(500000) < 0.5, 1, -1)))]
for name, ops in cases:
    packed, layout = clip_ops_scene(ops, np.random.default_rng(1)).resolve()
    eng.upload_scene(packed, layout)
    row = {"scene": name, "clips": int(layout.n_clips)}
    for label, seq in (("partitioned_us", False), ("one_wave_us", True)):
        eng.set_debug_flags(seq_clip=seq)
        eng.set_profiling([])
        for _ in range(3):
            eng.run_stages(256, 256, 0xFF000000, AaConfig.Area, 0, 3)
        eng.set_profiling(["clip"])
        before = eng.stage_ms()["clip"]
        for _ in range(10):
            eng.run_stages(256, 256, 0xFF000000, AaConfig.Area, 0, 3)
        after = eng.stage_ms()["clip"]
        row[label] = round(1e3 * (after[0] - before[0]) / max(after[1] - before[1], 1), 1)
    eng.set_debug_flags()
    print(json.dumps(row), flush=True)
