"""Static check of the gfx950 ISA of every kernel: global loads that are waited for (s_waitcnt vmcnt(0)) before the next
load is issued -- the signature of conditional loads in unrolled loops that the compiler serialises.
usage: python scripts/isa_load_chains.py   (compiles vello_amd/csrc/engine/*.hip with -save-temps into /tmp/isa)"""
import glob, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs("/tmp/isa", exist_ok=True)
for src in sorted(glob.glob(os.path.join(root, "vello_amd/csrc/engine/*.hip"))):
    name = os.path.basename(src)[:-4]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                    "-c", src, "-o", f"/tmp/isa/{name}.o", "-save-temps=obj"], cwd=os.path.join(root, "vello_amd/csrc"),
                   stderr=subprocess.DEVNULL)
    asm = f"/tmp/isa/{name}-hip-amdgcn-amd-amdhsa-gfx950.s"
    if not os.path.exists(asm):
        continue
    s = open(asm).read()
    for m in re.finditer(r'^(_ZN2vk\w+):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M):
        kname = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.split("(")[0]
        lines = [l.strip() for l in m.group(2).split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        loads = [i for i, l in enumerate(lines) if l.startswith(("global_load", "flat_load", "buffer_load"))]
        chains = 0
        for a, b in zip(loads, loads[1:]):
            between = lines[a + 1:b]
            if any(x.startswith("s_waitcnt vmcnt(0)") for x in between) and b - a < 40:
                chains += 1
        print(f"{kname:60s} instrs {len(lines):6d} loads {len(loads):4d} load->wait(0)->load within 40 instrs: {chains}")
