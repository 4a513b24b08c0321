"""Copies what scripts/gpu_profiles.sh left in gpurun_out/profiles_<tag>/ into profiles/ and stamps the commit that was
measured into pmc_traffic.json (the GPU box works on a snapshot without .git).  Run right after the gpurun call, before
any further commit:  python scripts/collect_profiles.py r02"""
import json
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(root, "gpurun_out", f"profiles_{tag}")
dst = os.path.join(root, "profiles")
commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"]).decode().strip()
dirty = subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--untracked-files=no"]).decode().strip()
for name in sorted(os.listdir(src)):
    if name in ("commit.txt",):
        continue
    if name == "pmc_traffic.json":
        d = json.load(open(os.path.join(src, name)))
        d["commit"] = commit + ("+uncommitted" if dirty else "")
        d["tag"] = tag
        json.dump(d, open(os.path.join(dst, name), "w"), indent=1)
    else:
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
print("profiles/ <-", tag, "at", commit, "(dirty)" if dirty else "")
