"""A/B of two builds of libvello_hip.so under bench.py.
   cp vello_amd/lib/libvello_hip.so ab_tmp/libvello_hip_B.so   (the other build; ab_tmp/ is git-ignored but travels with gpurun)
   python scripts/ab_bench.py A [bench args]    -> the in-tree library
   python scripts/ab_bench.py B [bench args]    -> ab_tmp/libvello_hip_B.so"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1]
import vello_amd._lib as L
if which != "A":
    L._use_library(os.path.join(ROOT, "ab_tmp", f"libvello_hip_{which}.so"))
sys.argv = ["bench.py"] + (sys.argv[2:] if len(sys.argv) > 2 else ["--steps", "200", "--warmup", "10", "--no-cpu-baseline"])
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
