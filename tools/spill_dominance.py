#!/usr/bin/env python3
"""Read-before-write check of spill slots in a gfx950 assembly listing (hipcc --cuda-device-only -S).

For every function: basic blocks and their successors are rebuilt from labels and branches, and a forward "must be written" data-flow
analysis runs over them: a register-allocator reload (`scratch_load_dword vN, off, s33|s32 offset:K ; 4-byte Folded Reload`) is REPORTED
when some path from the function entry reaches it without a spill store (`; 4-byte Folded Spill`) to the same slot.  Only the
allocator's own slots are checked (the compiler marks them): named stack objects are also written by callees through pointers,
which a per-function analysis cannot see (--all includes them: expect false positives).  The s_waitcnt side is checked too:
between a reload and the first use of its register there must be an s_waitcnt vmcnt that covers it (reloads complete in order).

    python tools/spill_dominance.py [--all] file.s [name-substring ...]

Used for the post-mortem of the round-1/2 wrong-lanes event: the failing build's listing (max-ilp + the experimental
pressure trackers, Fp4 squarings as calls) against the passing one of the same source."""
import re
import sys
from collections import OrderedDict

LOAD = re.compile(r"^scratch_load_(dword|dwordx2|dwordx3|dwordx4|ubyte|ushort|sbyte|sshort)\s+\S+,\s*off,\s*(s3[23]|off)(?:\s+offset:(-?\d+))?")
STORE = re.compile(r"^scratch_store_(dword|dwordx2|dwordx3|dwordx4|byte|short)\s+off,\s*\S+,\s*(s3[23]|off)(?:\s+offset:(-?\d+))?")
VLOAD = re.compile(r"^scratch_load_\w+\s+\S+,\s*v\d+")
VSTORE = re.compile(r"^scratch_store_\w+\s+v\d+")
SIZE = {"dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16, "ubyte": 1, "ushort": 2, "sbyte": 1, "sshort": 2, "byte": 1, "short": 2}
BRANCH = re.compile(r"^s_cbranch_\w+\s+(\S+)")
JUMP = re.compile(r"^s_branch\s+(\S+)")


def functions(path):
    funcs = OrderedDict()
    cur = None
    for line in open(path):
        m = re.match(r"^([A-Za-z_][\w.$]*):", line)
        if m and not m.group(1).startswith((".L", "BB", "Lfunc")):
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur is not None:
            funcs[cur].append(line.rstrip("\n"))
    return funcs


ONLY_SPILLS = True


def analyse(name, lines):
    # split into blocks
    blocks, labels, cur = [], {}, {"insts": [], "succ": [], "label": None}
    blocks.append(cur)
    sp_moves = 0
    for raw in lines:
        s = raw.strip()
        m = re.match(r"^(\.LBB[\w.]+):", s)
        if m:
            nxt = {"insts": [], "succ": [], "label": m.group(1)}
            if not cur.get("ends"):
                cur["succ"].append(nxt)
            blocks.append(nxt)
            labels[m.group(1)] = nxt
            cur = nxt
            continue
        if not s or s.startswith((".", ";", "//")):
            continue
        if ONLY_SPILLS and s.startswith("scratch_") and "Folded" not in s:
            s = "; " + s                                   # a named stack object: out of scope
            continue
        cur["insts"].append(s)
        if re.match(r"^s_add(k)?_i32 s3[23]\b|^s_mov(k)?_[ib]32 s3[23]\b|^s_sub_i32 s3[23]\b", s):
            sp_moves += 1
        mb, mj = BRANCH.match(s), JUMP.match(s)
        if mb or mj or s.startswith(("s_setpc_b64", "s_endpgm")):
            nxt = {"insts": [], "succ": [], "label": None}
            if mb:
                cur["jump_to"] = cur.get("jump_to", []) + [mb.group(1)]
                cur["succ"].append(nxt)
            elif mj:
                cur["jump_to"] = cur.get("jump_to", []) + [mj.group(1)]
                cur["ends"] = True
            else:
                cur["ends"] = True
            blocks.append(nxt)
            cur = nxt
    for b in blocks:
        for l in b.get("jump_to", []):
            if l in labels:
                b["succ"].append(labels[l])
    # data flow: set of written (base, byte) pairs that MUST be written on entry of each block
    idx = {id(b): i for i, b in enumerate(blocks)}
    preds = [[] for _ in blocks]
    for i, b in enumerate(blocks):
        for sct in b["succ"]:
            preds[idx[id(sct)]].append(i)
    def gen(b):
        w = set()
        for s in b["insts"]:
            m = STORE.match(s)
            if m:
                off = int(m.group(3) or 0)
                for k in range(SIZE[m.group(1)]):
                    w.add((m.group(2), off + k))
        return w
    gens = [gen(b) for b in blocks]
    ALL = None
    inn = [ALL] * len(blocks)
    inn[0] = set()
    changed = True
    out = [None] * len(blocks)
    while changed:
        changed = False
        for i, b in enumerate(blocks):
            if i:
                ps = [out[p] for p in preds[i] if out[p] is not None]
                if not ps:
                    continue
                new_in = set.intersection(*ps) if ps else set()
            else:
                new_in = set()
            new_out = new_in | gens[i]
            if inn[i] != new_in or out[i] != new_out:
                inn[i], out[i] = new_in, new_out
                changed = True
    # frame size: largest store offset + 16 (loads above it read the caller's outgoing arguments)
    frame = 0
    for g in gens:
        for (_, o) in g:
            frame = max(frame, o + 1)
    findings, nloads, nvec = [], 0, 0
    for i, b in enumerate(blocks):
        if inn[i] is None:
            continue
        w = set(inn[i])
        for s in b["insts"]:
            m = STORE.match(s)
            if m:
                off = int(m.group(3) or 0)
                for k in range(SIZE[m.group(1)]):
                    w.add((m.group(2), off + k))
                continue
            if VLOAD.match(s) or VSTORE.match(s):
                nvec += 1
                continue
            m = LOAD.match(s)
            if m:
                nloads += 1
                off = int(m.group(3) or 0)
                missing = [k for k in range(SIZE[m.group(1)]) if (m.group(2), off + k) not in w]
                if missing and off < frame and off >= 0:
                    findings.append((b["label"] or "entry", s))
    return nloads, nvec, sp_moves, frame, findings


VM_OP = re.compile(r"^(scratch|flat|global|buffer)_(load|store|atomic)")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def waitcnt_check(lines):
    """every register a reload writes must not be touched before an s_waitcnt vmcnt(N) that covers the reload (vector-memory operations
    of a wavefront complete in issue order on gfx9: vmcnt(N) means all but the youngest N are done).  Linear scan; at a label the
    state is kept (fall-through) — a branch target entered with a load in flight would only add reports, never hide one."""
    issued = 0
    pending = {}            # register -> sequence number of the reload that writes it
    bad = []
    for raw in lines:
        s = raw.split(";")[0].strip()
        if not s or s.startswith(".") or s.endswith(":"):
            continue
        if s.startswith("s_swappc_b64"):                     # a call: the callee's first instruction is s_waitcnt vmcnt(0) ... (checked in main)
            pending = {}
            continue
        m = re.match(r"^s_waitcnt\b(.*)", s)
        if m:
            v = re.search(r"vmcnt\((\d+)\)", m.group(1))
            if v:
                done = issued - int(v.group(1))
                pending = {r: q for r, q in pending.items() if q > done}
            continue
        if VM_OP.match(s):
            issued += 1
            ops = s.split(None, 1)[1] if " " in s else ""
            first = ops.split(",")[0]
            touched = regs_of(ops)
            if "Folded Reload" in raw and s.startswith("scratch_load"):
                dst = regs_of(first)
                hit = (touched - dst) & set(pending)
                if hit:
                    bad.append((raw.strip(), sorted(hit)))
                for r in dst:
                    pending[r] = issued
                continue
            hit = touched & set(pending)
            if hit:
                bad.append((raw.strip(), sorted(hit)))
            continue
        hit = regs_of(s) & set(pending)
        if hit:
            bad.append((raw.strip(), sorted(hit)))
            for r in hit:
                pending.pop(r, None)
    return bad


def flat_wait_check(lines):
    """flat_load results (the LDS slot through a generic pointer, private stack objects through a generic pointer): FLAT instructions may be
    served by LDS or by memory and return out of order with respect to each other, so a counter value other than zero proves nothing —
    the destination registers must not be touched before an s_waitcnt that brings BOTH vmcnt and lgkmcnt to 0 (or a call)."""
    pend, bad = set(), []
    for raw in lines:
        s = raw.split(";")[0].strip()
        if not s or s.startswith(".") or s.endswith(":"):
            continue
        if s.startswith("s_swappc_b64"):
            pend = set()
            continue
        if s.startswith("s_waitcnt"):
            # gfx9 syntax: fields that are not named stay at their maximum, i.e. are not waited for
            v, l = re.search(r"vmcnt\((\d+)\)", s), re.search(r"lgkmcnt\((\d+)\)", s)
            if v and l and int(v.group(1)) == 0 and int(l.group(1)) == 0:
                pend = set()
            continue
        if s.startswith("flat_load"):
            ops = s.split(None, 1)[1]
            dst = regs_of(ops.split(",")[0])
            hit = (regs_of(ops) - dst) & pend
            if hit:
                bad.append((raw.strip(), sorted(hit)))
            pend |= dst
            continue
        hit = regs_of(s) & pend
        if hit:
            bad.append((raw.strip(), sorted(hit)))
            pend -= hit
    return bad


def main():
    global ONLY_SPILLS
    args = sys.argv[1:]
    if args and args[0] == "--all":
        ONLY_SPILLS = False
        args = args[1:]
    path, filt = args[0], args[1:]
    total_f = total_l = 0
    bad = early = flat_bad = 0
    # every out-of-line function must start by waiting for the caller's loads (arguments may still be in flight at the call)
    nowait = []
    for name, lines in functions(path).items():
        body = [l.split(";")[0].strip() for l in lines]
        body = [l for l in body if l and not l.startswith(".") and not l.endswith(":")]
        is_kernel = any("s_endpgm" in l for l in body) and not any(l.startswith("s_setpc_b64") for l in body)
        if body and not is_kernel and any(l.startswith("s_setpc_b64") for l in body) and not (body[0].startswith("s_waitcnt") and "vmcnt(0)" in body[0]):
            nowait.append(name)
    print("out-of-line functions that do not start with s_waitcnt vmcnt(0): %d %s" % (len(nowait), nowait[:5]))
    for name, lines in functions(path).items():
        if filt and not any(f in name for f in filt):
            continue
        nloads, nvec, sp_moves, frame, findings = analyse(name, lines)
        if nloads == 0 and not findings:
            continue
        total_f += 1
        total_l += nloads
        tag = "OK " if not findings else "READ-BEFORE-WRITE"
        print("%-18s %-72s frame-relative loads %4d  vector-addressed scratch ops %4d  frame bytes written %5d" % (tag, name[:72], nloads, nvec, frame))
        for lab, s in findings[:12]:
            print("        %s: %s" % (lab, s))
        bad += len(findings)
        wb = waitcnt_check(lines)
        for text, regs in wb[:12]:
            print("        USED BEFORE ITS WAIT v%s: %s" % (regs, text))
        early += len(wb)
        fb = flat_wait_check(lines)
        for text, regs in fb[:12]:
            print("        FLAT RESULT TOUCHED BEFORE vmcnt(0) lgkmcnt(0) v%s: %s" % (regs, text))
        flat_bad += len(fb)
    print("functions with spill reloads: %d, frame-relative loads checked: %d, loads without a dominating store: %d, registers of a reload touched "
          "before a covering s_waitcnt: %d, flat-load results touched before a full wait: %d" % (total_f, total_l, bad, early, flat_bad))


if __name__ == "__main__":
    main()
