import sys, time, torch
sys.path.insert(0, '.')
import vello_amd, workloads
packed, layout = workloads.paris_like_scene().resolve()
for n in (1, 2, 3, 4):
    engines = [vello_amd.Engine(device=0) for _ in range(n)]
    frames = [torch.zeros((1600, 1600, 4), dtype=torch.uint8, device='cuda:0') for _ in range(n)]
    for e in engines: e.upload_scene(packed, layout)
    for it in range(10):
        for e, f in zip(engines, frames): e.render_resident(1600, 1600, 0xFFFFFFFF, 2, out=f)
    for e in engines: e.sync()
    K = 120
    t0 = time.perf_counter()
    for it in range(K // n):
        for e, f in zip(engines, frames): e.render_resident(1600, 1600, 0xFFFFFFFF, 2, out=f)
    for e in engines: e.sync()
    dt = time.perf_counter() - t0
    print(n, 'contexts:', round((K // n) * n / dt, 1), 'frames/s')
    same = all(torch.equal(frames[0], f) for f in frames)
    print('  frames identical:', same)
    del engines
