import json, os, sys, statistics
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
sys.argv=["x"]
import ab_contexts as R
import bench, torch
for key in ("d2","r1mix","mmark","tiger"):
    wl=bench.Workload(key,0)
    ring=[torch.zeros((wl.height,wl.width,4),dtype=torch.uint8,device="cuda:0") for _ in range(4)]
    torch.cuda.synchronize()
    e=R.make_engine("A", wl.caps); e.upload_scene(wl.packed, wl.layout)
    rows={}
    for rep in range(3):
        for setting in ("side_by_side","separate"):
            if setting=="separate": os.environ["VELLO_HIP_FLATTEN_SEPARATE"]="1"
            else: os.environ.pop("VELLO_HIP_FLATTEN_SEPARATE",None)
            r=R.measure(e,wl,ring); rows.setdefault(setting,[]).append(r)
    os.environ.pop("VELLO_HIP_FLATTEN_SEPARATE",None)
    for k,rs in rows.items():
        km={n:statistics.median([r["kernel_us"][n] for r in rs]) for n in rs[0]["kernel_us"]}
        print(key,k,"%.1f us"%statistics.median([r["latency_us"] for r in rs]),"flatten %.1f"%statistics.median([r["stage_us"]["flatten"] for r in rs]), km, flush=True)
