#!/usr/bin/env python3
"""Per-class instruction table of selected functions of a gfx950 listing (hipcc --cuda-device-only -S): what a routine issues besides
its multiply-adds.

    python tools/isa_classes.py file.s name-substring [...]

classes: mad (v_mad_i64_i32 / v_mad_u64_u32), mul_lo (v_mul_lo_u32, v_mul_hi), add64 (v_lshl_add_u64: the per-column join of two
multiply-add chains), shift64 (v_ashrrev_i64, v_lshrrev_b64, v_lshlrev_b64: the column carries), mask (v_and_b32 and friends),
add32 (v_add / v_sub / v_lshl_add_u32 / v_add3: lazy additions, carry rounds), shift32, select (v_cndmask), move (v_mov, incl. DPP),
xlane (ds_bpermute / ds_swizzle / v_readlane ...), lds (ds_read / ds_write), scratch, flat (flat / global), salu, wait (s_waitcnt, s_nop)."""
import re
import sys
from collections import Counter, OrderedDict

CLASSES = ["mad", "mul_lo", "add64", "shift64", "mask", "add32", "shift32", "select", "move", "other_valu", "xlane", "lds", "scratch", "flat", "salu", "wait"]


def classify(op):
    if op.startswith(("v_mad_i64_i32", "v_mad_u64_u32")):
        return "mad"
    if op.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u32", "v_mad_i32", "v_mul_u32", "v_mul_i32")):
        return "mul_lo"
    if op.startswith("v_lshl_add_u64"):
        return "add64"
    if op.startswith(("v_ashrrev_i64", "v_lshrrev_b64", "v_lshlrev_b64")):
        return "shift64"
    if op.startswith(("v_and_", "v_or_", "v_or3", "v_xor", "v_and_or", "v_bfe", "v_bfi", "v_not")):
        return "mask"
    if op.startswith(("v_add", "v_sub", "v_lshl_add_u32", "v_addc", "v_subb", "v_max", "v_min")):
        return "add32"
    if op.startswith(("v_ashrrev_i32", "v_lshrrev_b32", "v_lshlrev_b32", "v_alignbit", "v_lshl_or")):
        return "shift32"
    if op.startswith(("v_cndmask", "v_cmp")):
        return "select"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "move"
    if op.startswith(("ds_bpermute", "ds_permute", "ds_swizzle", "v_readlane", "v_readfirstlane", "v_writelane")):
        return "xlane"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("flat_", "global_", "buffer_")):
        return "flat"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "other_valu"
    return None


def main():
    path, names = sys.argv[1], sys.argv[2:]
    funcs, cur = OrderedDict(), None
    for line in open(path):
        m = re.match(r"^([A-Za-z_][\w.$]*):", line)
        if m and not m.group(1).startswith((".L", "BB", "Lfunc")):
            cur = m.group(1)
            funcs[cur] = Counter()
            continue
        if cur is None:
            continue
        s = line.strip()
        if not s or s.startswith((".", ";", "//")):
            continue
        c = classify(s.split()[0])
        if c:
            funcs[cur][c] += 1
    print("%-34s %7s" % ("function", "total") + "".join(" %7s" % c[:7] for c in CLASSES) + "  non-mad VALU / mad")
    for n in names:
        for k, c in funcs.items():
            if n in k:
                tot = sum(c.values())
                valu = sum(c[x] for x in ("mul_lo", "add64", "shift64", "mask", "add32", "shift32", "select", "move", "other_valu"))
                short = re.sub(r"^_ZN6c12381\d+", "", k)[:34]
                print("%-34s %7d" % (short, tot) + "".join(" %7d" % c[x] for x in CLASSES) + "  %.3f" % (valu / max(c["mad"], 1)))


if __name__ == "__main__":
    main()
