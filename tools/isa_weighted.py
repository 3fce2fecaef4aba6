#!/usr/bin/env python3
"""Instruction classes of one unit of work from a gfx950 listing: the per-routine class counts of tools/isa_classes.py weighted by the calls each
out-of-line routine gets per unit (the three-lane pairing's routines are real calls with fixed trip counts), so that the lane-instructions of a
pairing that are NOT multiply-adds have owners.

    python tools/isa_weighted.py listing.s name=calls [name=calls ...]      (name = substring of the mangled function name)

Counts are static instructions of each routine (one wavefront executes every one of them per call: the routines are branch-free apart from
wave-uniform skips noted in the output), per LANE; a pairing occupies 3 lanes of a 63-lane group (64 / 21 = 3.05 lanes with the idle one)."""
import re
import sys
from collections import Counter, OrderedDict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from isa_classes import CLASSES, classify  # noqa: E402


def parse(path):
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
    return funcs


def main():
    funcs = parse(sys.argv[1])
    rows, total = [], Counter()
    for spec in sys.argv[2:]:
        name, calls = spec.rsplit("=", 1)
        calls = float(calls)
        hit = [k for k in funcs if name in k]
        if len(hit) != 1:
            raise SystemExit("%s matches %d functions: %s" % (name, len(hit), hit[:5]))
        c = funcs[hit[0]]
        rows.append((name, calls, c))
        for k, v in c.items():
            total[k] += v * calls
    valu_other = ("mul_lo", "add64", "shift64", "mask", "add32", "shift32", "select", "move", "other_valu")
    print("%-26s %6s %8s" % ("routine", "calls", "instr") + "".join(" %7s" % c[:7] for c in CLASSES) + "  other VALU / mad   share of the unit")
    grand = sum(total.values())
    for name, calls, c in rows:
        tot = sum(c.values())
        ov = sum(c[x] for x in valu_other)
        print("%-26s %6g %8d" % (name[:26], calls, tot) + "".join(" %7d" % c[x] for x in CLASSES) + "  %.3f              %5.1f %%" % (ov / max(c["mad"], 1), 100 * tot * calls / grand))
    print("%-26s %6s %8d" % ("per lane and unit", "", grand) + "".join(" %7d" % total[x] for x in CLASSES) + "  %.3f" % (sum(total[x] for x in valu_other) / max(total["mad"], 1)))
    print("%-26s %6s %8s" % ("share of the stream", "", "") + "".join(" %6.1f%%" % (100 * total[x] / grand) for x in CLASSES))
    vec = sum(total[x] for x in ("mad",) + valu_other)
    print("vector ALU instructions per lane and unit: %d (%.1f %% multiply-adds); x 64 / 21 lanes per pairing = %.3f M lane-instructions" % (vec, 100 * total["mad"] / vec, vec * 64 / 21 / 1e6))
    nonmad = vec - total["mad"]
    print("the %d non-multiply-add vector instructions by class: " % nonmad + ", ".join("%s %.1f %%" % (x, 100 * total[x] / nonmad) for x in valu_other if total[x]))


if __name__ == "__main__":
    main()
