#!/usr/bin/env python3
"""DESIGN.md 3.0's table from the evidence files of a round (generated, not remembered):
    python scripts/stage_table.py r06 [d2]
reads profiles/<tag>_bench.json (stage_algorithmic_bytes), profiles/<tag>_kernel_stats_serial_<scene>.csv (rocprofv3, one frame at a
time; the create-time warm-up launches are the 'Calls' beyond the timed ones and carry no weight in the means of 60+ calls),
profiles/pmc_traffic.json (FETCH_SIZE x 2 + WRITE_SIZE per kernel) and prints the stage table in markdown."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 8000.0  # GB/s


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    scene = sys.argv[2] if len(sys.argv) > 2 else "d2"
    P = lambda n: os.path.join(ROOT, "profiles", n)
    bench = json.load(open(P(f"{tag}_bench.json")))
    alg = bench["roofline"]["stage_algorithmic_bytes"]
    traffic = json.load(open(P("pmc_traffic.json")))
    tr = traffic["workloads"][scene]["kernels"]
    us = {}
    for r in csv.DictReader(open(P(f"{tag}_kernel_stats_serial_{scene}.csv"))):
        n = r["Name"]
        for k in ("k_pathtag_scan", "k_flatten_light", "k_flatten_main", "k_flatten_tail", "k_binning_tile_alloc", "k_path_count", "k_backdrop", "k_coarse_prep", "k_coarse(",
                  "k_path_tiling", "k_fine<2, false, 4>"):
            if k in n and int(r["Calls"]) >= 20:
                us[k] = float(r["AverageNs"]) / 1e3
    stages = [("pathtag_scan", ["k_pathtag_scan"], ["pathtag_scan"], ["k_pathtag_scan"]),
              ("flatten (light + main + tail)", ["k_flatten_light", "k_flatten_main", "k_flatten_tail"], ["flatten"], ["k_flatten_light", "k_flatten_main", "k_flatten_tail"]),
              ("binning + tile_alloc (one launch)", ["k_binning_tile_alloc"], ["binning", "tile_alloc"], ["k_binning_tile_alloc"]),
              ("path_count", ["k_path_count"], ["path_count"], ["k_path_count"]),
              ("backdrop", ["k_backdrop"], ["backdrop"], ["k_backdrop"]),
              ("coarse (prep + `k_coarse`)", ["k_coarse_prep", "k_coarse("], ["coarse"], ["k_coarse_prep", "k_coarse"]),
              ("path_tiling", ["k_path_tiling"], ["path_tiling"], ["k_path_tiling"]),
              ("**fine (`k_fine<2,false,4>`, dominant)**", ["k_fine<2, false, 4>"], ["fine"], ["k_fine"])]
    print(f"(commit {bench['config'].get('commit')}, traffic at {traffic.get('commit')}; {scene}, one frame at a time)")
    print("| stage | µs | algorithmic MB | GB/s | frac of 8 TB/s | PMC traffic / algorithmic |")
    print("|---|---|---|---|---|---|")
    t_us = t_a = t_t = 0.0
    for name, ks, algs, trs in stages:
        u = sum(us[k] for k in ks)
        a = sum(alg[x] for x in algs) / 1e6
        t = sum(tr[k]["traffic_bytes"] for k in trs) / 1e6
        t_us += u; t_a += a; t_t += t
        print(f"| {name} | {u:.1f} | {a:.1f} | {a / u * 1e3:.0f} | {a / u * 1e3 / PEAK:.3f} | {t / a:.2f} ({t:.1f} MB) |")
    print(f"| sum | {t_us:.0f} | {t_a:.0f} | {t_a / t_us * 1e3:.0f} | {t_a / t_us * 1e3 / PEAK:.3f} | {t_t / t_a:.2f} ({t_t:.0f} MB) |")


if __name__ == "__main__":
    main()
