#!/usr/bin/env python3
"""Basic-block instruction census of one kernel in a hipcc -S dump.

    hipcc --offload-arch=gfx950 ... -S --cuda-device-only -o fine.s engine/fine.hip
    scripts/isa_blocks.py fine.s k_fineILi2ELb0E [--dump LABEL]

Prints every basic block (label .. next label) with its VALU / SALU / LDS / VMEM / other counts and the labels it
branches to, marking backward branches (loops).  What k_fine's "instructions per fill" are made of is read off the
blocks of the fill loop; PMC counts say how many, this says which."""
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
            return "wait"
        if op.startswith(("s_load", "s_buffer_load")):
            return "smem"
        return "salu"
    return "other"


# issue cost in cycles per wave64 instruction and SIMD, measured on MI355X with scripts/calib/valu_rate.hip
# (profiles/r04_valu_rate.txt): the plain VOP1/VOP2 integer / logic / f32 forms take ~2.3, everything VOP3-only, shifts
# left, compares, conversions, DPP / SDWA forms, packed f32, 24- and 32-bit multiplies ~4.3, rcp and friends ~8.3
FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_mov_b32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_cndmask_b32_e32",
        "v_bitop3_b32", "v_not_b32", "v_add_co_u32", "v_accvgpr")
SLOW8 = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def valu_cycles(text):
    op = text.split()[0]
    if "dpp" in text or "sdwa" in text:
        return 4.3
    if op.startswith(SLOW8):
        return 8.3
    if op == "v_readlane_b32" and re.search(r", s\d+$", text.strip()):
        return 8.3
    if op.endswith("_e64"):
        return 4.3
    base = op[:-4] if op.endswith("_e32") else op
    if op == "v_cndmask_b32_e32" or base in FAST or op in FAST:
        # a VOP2 with an SGPR source operand measured at the slow rate
        if re.search(r"[ ,]s\d+|s\[\d+:\d+\]", text.split(None, 1)[1] if " " in text else "") and op != "v_cndmask_b32_e32":
            return 4.3
        return 2.3
    return 4.3


def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    # the function may have several s_endpgm; take the .Lfunc_end
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks = []
    cur = {"label": "entry", "line": start, "counts": {}, "targets": [], "text": []}
    order = {}
    for i in range(start + 1, end):
        l = lines[i]
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "line": i, "counts": {}, "targets": [], "text": []}
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        c = classify(op)
        cur["counts"][c] = cur["counts"].get(c, 0) + 1
        if c == "valu":
            cur["counts"]["cyc"] = cur["counts"].get("cyc", 0.0) + valu_cycles(s)
        cur["text"].append(s)
        if op.startswith(("s_cbranch", "s_branch")):
            cur["targets"].append(s.split()[1])
    blocks.append(cur)
    for n, b in enumerate(blocks):
        order[b["label"]] = n
    tot = {}
    for n, b in enumerate(blocks):
        if dump is not None:
            if b["label"] == dump:
                print("\n".join(b["text"]))
            continue
        c = b["counts"]
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
        tg = ["%s%s" % (t, "^" if order.get(t, 1 << 30) <= n else "") for t in b["targets"]]
        print("%-12s valu %4d (%5.0f cyc) salu %4d lds %3d vmem %3d smem %2d wait %3d  -> %s" % (
            b["label"], c.get("valu", 0), c.get("cyc", 0.0), c.get("salu", 0), c.get("lds", 0), c.get("vmem", 0), c.get("smem", 0), c.get("wait", 0), " ".join(tg)))
    if dump is None:
        print("total", tot)


if __name__ == "__main__":
    main()
