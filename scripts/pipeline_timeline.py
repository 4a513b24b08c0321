"""What overlaps with what when frames are in flight: reads rocprofv3's kernel trace (start / end time stamps per dispatch) of
`bench.py --timed-only --in-flight N` and prints, for the steady-state window, every kernel's mean duration under overlap, the
chip's concurrency histogram (how long 0, 1, 2, ... kernels were running at once) and, per kernel, the mean number of OTHER
kernels running beside it.    python scripts/pipeline_timeline.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void vk::", "").replace("vk::", "")
    return n.split("(")[0][:28]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = []
    for r in rows:
        name = r.get("Kernel_Name") or r.get("Name")
        if "vk::" not in name:
            continue
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(name), r.get("Queue_Id", "?")))
    ev.sort()
    if not ev:
        print("no engine kernels in the trace")
        return
    # steady state: the middle 60 % of the k_fine launches
    fines = [e for e in ev if e[2].startswith("k_fine")]
    lo, hi = fines[len(fines) // 5][0], fines[len(fines) * 4 // 5][1]
    win = [e for e in ev if e[0] >= lo and e[1] <= hi]
    n_frames = sum(1 for e in win if e[2].startswith("k_fine"))
    print(f"window {(hi - lo) / 1e3:.0f} us, {n_frames} frames -> {(hi - lo) / 1e3 / max(n_frames, 1):.1f} us per frame; queues: {sorted(set(e[3] for e in win))}")
    # concurrency histogram
    pts = []
    for s, e, n, q in win:
        pts.append((s, 1, n))
        pts.append((e, -1, n))
    pts.sort()
    hist = defaultdict(int)
    running = defaultdict(int)
    beside = defaultdict(float)   # kernel -> integral of (others running) over its own run time
    own = defaultdict(float)
    cur, t_prev = 0, pts[0][0]
    for t, d, n in pts:
        dt = t - t_prev
        if dt > 0:
            hist[cur] += dt
            for k, c in running.items():
                if c > 0:
                    own[k] += dt * c
                    beside[k] += dt * c * (cur - 1)
        t_prev = t
        cur += d
        running[n] += d
    # who runs alone, and how busy each queue is
    alone = defaultdict(int)
    cur_set = defaultdict(int)
    t_prev = pts[0][0]
    for t, d, n in pts:
        act = [k for k, c in cur_set.items() if c > 0]
        if sum(cur_set.values()) == 1 and t > t_prev:
            alone[act[0]] += t - t_prev
        t_prev = t
        cur_set[n] += d
    qbusy = defaultdict(int)
    for s0, e0, n, q in win:
        qbusy[q] += e0 - s0
    tot = sum(hist.values())
    print("alone on the chip:", "  ".join(f"{k} {v / tot * 100:.0f} %" for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:6]))
    print("queue busy (a kernel of it executing):", "  ".join(f"q{q} {v / (hi - lo) * 100:.0f} %" for q, v in sorted(qbusy.items())))
    print("kernels running at once:", "  ".join(f"{k}: {v / tot * 100:.0f} %" for k, v in sorted(hist.items())))
    dur = defaultdict(list)
    for s, e, n, q in win:
        dur[n].append((e - s) / 1e3)
    print(f"{'kernel':28s} {'launches':>8s} {'mean us':>8s} {'us/frame':>9s} {'others beside it':>17s}")
    for n in sorted(dur, key=lambda k: -sum(dur[k])):
        print(f"{n:28s} {len(dur[n]):8d} {sum(dur[n]) / len(dur[n]):8.1f} {sum(dur[n]) / max(n_frames, 1):9.1f} {beside[n] / max(own[n], 1):17.2f}")
    print(f"sum of kernel time per frame: {sum(sum(v) for v in dur.values()) / max(n_frames, 1):.0f} us")
    # the gaps inside a queue: from the end of a kernel to the start of the next one of the same queue (round 6)
    byq = defaultdict(list)
    for s0, e0, n, q in win:
        byq[q].append((s0, e0, n))
    gaps = defaultdict(list)
    for q, lst in byq.items():
        lst.sort()
        for (s0, e0, n0), (s1, e1, n1) in zip(lst[:-1], lst[1:]):
            gaps[(n0, n1)].append((s1 - e0) / 1e3)
    print(f"{'gap behind':28s} {'before':28s} {'n':>4s} {'mean us':>8s} {'median':>8s} {'us/frame':>9s}")
    tot_gap = 0.0
    for (n0, n1), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        tot_gap += sum(v)
        if len(v) >= 4:
            print(f"{n0:28s} {n1:28s} {len(v):4d} {sum(v) / len(v):8.1f} {v2[len(v2) // 2]:8.1f} {sum(v) / max(n_frames, 1):9.1f}")
    print(f"sum of the gaps per frame: {tot_gap / max(n_frames, 1):.0f} us")


if __name__ == "__main__":
    main()
