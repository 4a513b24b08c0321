"""Turns the FETCH_SIZE / WRITE_SIZE summaries of scripts/gpu_profiles.sh into profiles/pmc_traffic.json: bytes per
launch and kernel, corrected with the factors the calibration kernels (scripts/calib/pmc_calib.hip, known 512 MiB each)
give in the same session, stamped with the commit that was measured."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vello_amd._lib import kernel_sources_hash  # noqa: E402

out_dir, tag = sys.argv[1], sys.argv[2]


def parse(path):
    res, cur = {}, None
    if not os.path.exists(path):
        return res
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip().split("(")[0].replace("void ", "").replace("vk::", "")
            cur = re.sub(r"<.*", "", cur)
            res.setdefault(cur, {})
        else:
            m = re.match(r"\s+(\S+)\s+avg/dispatch\s+([0-9.]+)", line)
            if m and cur:
                res[cur][m.group(1)] = res[cur].get(m.group(1), 0.0) + float(m.group(2))
    return res


KNOWN = 536870912
cal_f = parse(os.path.join(out_dir, f"{tag}_pmc_calib_fetch.summary.txt"))
cal_w = parse(os.path.join(out_dir, f"{tag}_pmc_calib_write.summary.txt"))
fetch_corr = write_corr = None
if "calib_read16" in cal_f and cal_f["calib_read16"].get("FETCH_SIZE"):
    fetch_corr = KNOWN / (cal_f["calib_read16"]["FETCH_SIZE"] * 1024.0)
if "calib_write16" in cal_w and cal_w["calib_write16"].get("WRITE_SIZE"):
    write_corr = KNOWN / (cal_w["calib_write16"]["WRITE_SIZE"] * 1024.0)
fc = round(fetch_corr, 3) if fetch_corr else 2.0
wc = round(write_corr, 3) if write_corr else 1.0
commit = None
try:
    commit = open(os.path.join(out_dir, "commit.txt")).read().strip() or None
except OSError:
    pass
doc = {
    "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE collected in separate passes (kernel trace only) of `python bench.py --workload W "
            "--steps 8 --warmup 2 --in-flight 1 --timed-only`; counters are in KiB, averaged per dispatch.  Corrections from the calibration "
            "kernels of the same session (scripts/calib/pmc_calib.hip, 512 MiB each): streamed 16 B/lane reads and writes.",
    "commit": commit,
    "kernel_sources": kernel_sources_hash(),  # (bench.py compares it with the sources it runs: roofline.traffic_stale)
    "fetch_correction": fc,
    "write_correction": wc,
    "calibration": {k: {"FETCH_SIZE_KiB": cal_f.get(k, {}).get("FETCH_SIZE"), "WRITE_SIZE_KiB": cal_w.get(k, {}).get("WRITE_SIZE"), "known_bytes": KNOWN}
                    for k in sorted(set(cal_f) | set(cal_w)) if k.startswith("calib_")},
    "workloads": {},
}
for wl in ("d2", "r1mix"):
    f = parse(os.path.join(out_dir, f"{tag}_pmc_fetch_{wl}.summary.txt"))
    w = parse(os.path.join(out_dir, f"{tag}_pmc_write_{wl}.summary.txt"))
    kernels = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        fb = f.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0 * fc
        wb = w.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0 * wc
        kernels[k] = {"fetch_bytes": int(fb), "write_bytes": int(wb), "traffic_bytes": int(fb + wb)}
    doc["workloads"][wl] = {"kernels": kernels}
print(json.dumps(doc, indent=1))
