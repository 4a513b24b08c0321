"""One frame at a time, kernel by kernel: reads rocprofv3's kernel trace (start / end time stamps per dispatch) of frames rendered
one after the other (scripts/small_scene_trace.py) and prints, as medians over the steady-state frames, every launch of a frame:
when it starts after the frame's first launch, how long it runs, and the gap since the launch before it ended -- then the frame's
span, the sum of its kernels and the sum of its gaps, and the idle time between frames (host: wait, enqueue).
    python scripts/frame_timeline.py <kernel_trace.csv>"""
import csv
import sys
from statistics import median


def short(n):
    n = n.replace("void vk::", "").replace("vk::", "")
    return n.split("(")[0][:34]


def main():
    ev = []
    for r in csv.DictReader(open(sys.argv[1])):
        name = r.get("Kernel_Name") or r.get("Name")
        if "vk::" not in name and "fill" not in name.lower() and "memset" not in name.lower():
            continue
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(name)))
    ev.sort()
    frames, cur = [], []
    for e in ev:  # a frame ends with its k_fine
        cur.append(e)
        if e[2].startswith("k_fine"):
            frames.append(cur)
            cur = []
    frames = frames[len(frames) // 4: len(frames) * 3 // 4]
    shape = [tuple(e[2] for e in f) for f in frames]
    common = max(set(shape), key=shape.count)
    frames = [f for f, s in zip(frames, shape) if s == common]
    print(f"{len(frames)} frames of {len(common)} launches")
    print(f"{'launch':36s} {'start':>8s} {'runs':>8s} {'gap before':>10s}   (us, medians)")
    for i, n in enumerate(common):
        st = median(f[i][0] - f[0][0] for f in frames) / 1e3
        du = median(f[i][1] - f[i][0] for f in frames) / 1e3
        gp = median(f[i][0] - f[i - 1][1] for f in frames) / 1e3 if i else 0.0
        print(f"{n:36s} {st:8.1f} {du:8.1f} {gp:10.1f}")
    span = median(f[-1][1] - f[0][0] for f in frames) / 1e3
    ksum = median(sum(e[1] - e[0] for e in f) for f in frames) / 1e3
    between = median(b[0][0] - a[-1][1] for a, b in zip(frames, frames[1:])) / 1e3 if len(frames) > 1 else 0.0
    print(f"frame: first start -> last end {span:.1f} us; kernels {ksum:.1f} us; gaps {span - ksum:.1f} us; between frames {between:.1f} us")


if __name__ == "__main__":
    main()
