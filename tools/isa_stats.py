"""Static instruction mix per function of a gfx950 assembly listing (hipcc --cuda-device-only -S).

usage: python tools/isa_stats.py k_pair3.s [name-substring ...]
Columns: mad = v_mad_[iu]64_[iu]32, valu = other v_* (without v_accvgpr), acc = v_accvgpr_*, scr = scratch_*, ds = ds_*,
flat = flat_*/global_*/buffer_*, salu = s_* (without s_waitcnt/s_nop), wait = s_waitcnt + s_nop."""
import re
import sys
from collections import OrderedDict


def main():
    path = sys.argv[1]
    filt = sys.argv[2:]
    funcs = OrderedDict()
    cur = None
    for line in open(path):
        m = re.match(r"^([A-Za-z_][\w.$]*):", line)
        if m and not m.group(1).startswith((".L", "BB", "Lfunc")):
            cur = m.group(1)
            funcs[cur] = dict(mad=0, valu=0, acc=0, scr=0, ds=0, flat=0, salu=0, wait=0, call=0)
            continue
        if cur is None:
            continue
        s = line.strip()
        if not s or s.startswith((".", ";", "//")):
            continue
        op = s.split()[0]
        d = funcs[cur]
        if op.startswith("v_mad_i64_i32") or op.startswith("v_mad_u64_u32"):
            d["mad"] += 1
        elif op.startswith("v_accvgpr"):
            d["acc"] += 1
        elif op.startswith("v_"):
            d["valu"] += 1
        elif op.startswith("scratch_"):
            d["scr"] += 1
        elif op.startswith("ds_"):
            d["ds"] += 1
        elif op.startswith(("flat_", "global_", "buffer_")):
            d["flat"] += 1
        elif op.startswith(("s_waitcnt", "s_nop")):
            d["wait"] += 1
        elif op.startswith("s_swappc") or op.startswith("s_setpc"):
            d["call"] += 1
        elif op.startswith("s_"):
            d["salu"] += 1
    print("%-70s %7s %7s %6s %6s %6s %6s %6s %6s %5s" % ("function", "mad", "valu", "acc", "scr", "ds", "flat", "salu", "wait", "call"))
    for name, d in funcs.items():
        if sum(d.values()) < 20:
            continue
        if filt and not any(f in name for f in filt):
            continue
        print("%-70s %7d %7d %6d %6d %6d %6d %6d %6d %5d" % (name[:70], d["mad"], d["valu"], d["acc"], d["scr"], d["ds"], d["flat"], d["salu"], d["wait"], d["call"]))


if __name__ == "__main__":
    main()
