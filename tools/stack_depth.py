"""Check a gfx950 assembly listing (hipcc --cuda-device-only -S): for every kernel, the deepest chain of stack frames
(per-lane bytes: kernel frame + frames of the out-of-line routines along every call path) against the
.private_segment_fixed_size the kernel descriptor declares.  A kernel whose chain is deeper than its declaration
lets one wavefront's stack run into the scratch of the next one.
usage: python tools/stack_depth.py file.s"""
import re
import sys
from collections import OrderedDict


def main():
    path = sys.argv[1]
    funcs = OrderedDict()
    cur = None
    for line in open(path):
        m = re.match(r"^([A-Za-z_][\w.$]*):", line)
        if m and not m.group(1).startswith((".L", "BB", "Lfunc")):
            cur = m.group(1)
            funcs[cur] = {"frame": 0, "calls": set(), "init_sp": None, "max_off": 0}
            continue
        if cur is None:
            continue
        s = line.strip()
        f = funcs[cur]
        m = re.match(r"s_addk_i32 s32, (0x[0-9a-f]+|\d+)", s) or re.match(r"s_add_i32 s32, s32, (0x[0-9a-f]+|\d+)", s)
        if m:
            v = int(m.group(1), 0)
            if v < 0x8000:
                f["frame"] = max(f["frame"], v)
        m = re.match(r"s_movk_i32 s32, (0x[0-9a-f]+|\d+)", s) or re.match(r"s_mov_b32 s32, (0x[0-9a-f]+|\d+)", s)
        if m and f["init_sp"] is None:
            f["init_sp"] = int(m.group(1), 0)
        m = re.search(r"([A-Za-z_][\w.$]*)@rel32@lo", s)
        if m:
            f["calls"].add(m.group(1))
        m = re.search(r"scratch_(?:load|store)\w* .*s33 offset:(\d+)", s)
        if m:
            f["max_off"] = max(f["max_off"], int(m.group(1)) + 16)
    declared = {}
    name = None
    for line in open(path):
        m = re.match(r"\s*\.amdhsa_kernel (\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"\s*\.amdhsa_private_segment_fixed_size (\d+)", line)
        if m and name:
            declared[name] = int(m.group(1))
    memo = {}

    def depth(fn, seen=()):
        if fn in memo:
            return memo[fn]
        if fn not in funcs or fn in seen:
            return 0
        f = funcs[fn]
        own = max(f["frame"], f["max_off"])
        d = own + max([depth(c, seen + (fn,)) for c in f["calls"] if c in funcs] + [0])
        memo[fn] = d
        return d

    print("%-90s %8s %8s %s" % ("kernel", "declared", "deepest", ""))
    bad = 0
    for k, decl in declared.items():
        f = funcs.get(k)
        if not f:
            continue
        own = f["init_sp"] or 0
        d = own + max([depth(c) for c in f["calls"] if c in funcs] + [0])
        flag = "OVERRUN by %d" % (d - decl) if d > decl else ""
        bad += d > decl
        print("%-90s %8d %8d %s" % (k[:90], decl, d, flag))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
