#!/usr/bin/env python3
"""Register / spill / scratch / LDS table of every kernel in the BUILT library, read from the code objects' metadata
(llvm-objdump --offloading + llvm-readelf --notes): evidence generated from the binary, not remembered.

    python scripts/kernel_resources.py [path/to/libvello_hip.so] > profiles/r06_kernel_resources.txt

Waves per SIMD follow from the VGPR count (512 registers per SIMD lane on gfx950, granule 8) and, per CU, from the LDS
(160 KB).  DESIGN.md section 3 quotes this file."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    short = []
    for d in out[: len(names)]:
        d = re.sub(r"^void ", "", d)
        d = d.split("(")[0].replace("vk::", "")
        d = d.replace("(anonymous namespace)::", "")
        short.append(d)
    return short


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "vello_amd", "lib", "libvello_hip.so")
    tmp = tempfile.mkdtemp(prefix="vello_kr_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, capture_output=True, check=True)
        rows = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], cwd=tmp, capture_output=True, text=True).stdout
            cur = {}
            for line in notes.split("\n"):
                m = re.match(r"\s+- \.agpr_count:|\s+- \.args:", line)
                if m and cur.get("name"):
                    rows.append(cur)
                    cur = {}
                m = re.match(r"\s+-?\s*\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size|agpr_count|max_flat_workgroup_size):\s+(\S+)", line)
                if m:
                    if m.group(1) == "name" and "name" in cur:
                        continue
                    cur[m.group(1)] = m.group(2)
            if cur.get("name"):
                rows.append(cur)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rows = [r for r in rows if "vgpr_count" in r]
    names = demangle([r["name"] for r in rows])
    stamp = ""
    # (the GPU box holds a snapshot without .git: scripts/grun.sh leaves the commit in .commit_stamp)
    if os.path.isdir(os.path.join(ROOT, ".git")):
        try:
            stamp = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
            if subprocess.run(["git", "-C", ROOT, "diff", "--quiet", "HEAD", "--", "vello_amd/csrc", "include"], capture_output=True).returncode != 0:
                stamp += "-dirty"
        except Exception:
            pass
    elif os.path.exists(os.path.join(ROOT, ".commit_stamp")):
        stamp = open(os.path.join(ROOT, ".commit_stamp")).read().strip()
    print(f"# {os.path.relpath(lib, ROOT)} built from {stamp}: every kernel's resources from the code objects' metadata (scripts/kernel_resources.py)")
    print(f"# waves/SIMD = min(8, floor(512 / VGPRs rounded up to 8)); LDS workgroups/CU = floor(160 KB / LDS)")
    print(f"{'kernel':48s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'v-spill':>7s} {'s-spill':>7s} {'scratch B':>9s} {'LDS B':>7s} {'waves/SIMD':>10s} {'LDS wg/CU':>9s}")
    seen = set()
    for r, n in sorted(zip(rows, names), key=lambda x: x[1]):
        if n in seen:
            continue
        seen.add(n)
        v = int(r["vgpr_count"]) + int(r.get("agpr_count", 0))
        vg = (v + 7) // 8 * 8
        wps = min(8, 512 // max(vg, 8))
        lds = int(r.get("group_segment_fixed_size", 0))
        wg = "-" if lds == 0 else str(160 * 1024 // lds)
        print(f"{n[:48]:48s} {r['vgpr_count']:>5s} {r.get('agpr_count', '0'):>5s} {r['sgpr_count']:>5s} {r.get('vgpr_spill_count', '0'):>7s} "
              f"{r.get('sgpr_spill_count', '0'):>7s} {r.get('private_segment_fixed_size', '0'):>9s} {lds:>7d} {wps:>10d} {wg:>9s}")
    print("# (kernels with dynamic LDS -- k_coarse, the flatten kernels -- add what engine.hip passes at launch: DESIGN.md 3.3 / 3.4)")


if __name__ == "__main__":
    main()
