"""N frames of one bench workload, one frame at a time -- the process rocprofv3 wraps for per-kernel statistics of a BASELINE config other than
the headline's:    rocprofv3 --kernel-trace --stats ... -- python scripts/render_loop.py tiger|mmark|d2|r1mix [frames]"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vello_amd  # noqa: E402

wl = bench.Workload(sys.argv[1], 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
eng = vello_amd.Engine(capacities=wl.caps) if wl.caps else vello_amd.Engine()
eng.upload_scene(wl.packed, wl.layout)
out = torch.zeros((wl.height, wl.width, 4), dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
for _ in range(n):
    eng.render_resident(wl.width, wl.height, bench.BASE_COLOR, wl.aa, out=out)
    eng.sync_frame(0)
assert eng.sync() == 0, eng.bump()
print(sys.argv[1], n, "frames", eng.bump())
