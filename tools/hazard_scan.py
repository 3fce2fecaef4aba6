#!/usr/bin/env python3
"""Static scan of a `hipcc -S` listing for the gfx9-family data hazards that need software wait states (the ones the hazard recognizer
is supposed to pad with s_nop) — a post-mortem aid for the wrong-lanes event: two builds of the same source that differ only
in instruction order are compared pattern by pattern.  No GPU involved.

Patterns (producer -> consumer: required wait states; AMD GCN3 / CDNA3 ISA guides, "Manually inserted wait states", and LLVM's
GCNHazardRecognizer for gfx940/gfx950):
  valu_sgpr_vmem        VALU writes an SGPR            -> VMEM / flat / scratch / buffer instruction reads that SGPR          5
  valu_sgpr_lane        VALU writes an SGPR / VCC      -> v_readlane / v_writelane uses it as the lane select                  4
  valu_vcc_divfmas      VALU writes VCC                -> v_div_fmas                                                            4
  valu_exec_dpp         VALU writes EXEC               -> VALU with a DPP modifier                                              5
  valu_vgpr_dpp         VALU writes a VGPR             -> VALU DPP reads that VGPR                                              2
  salu_m0_lds           SALU writes M0                 -> LDS "add-TID" / GDS / s_sendmsg                                        1
  trans_use             VALU transcendental result     -> non-transcendental VALU reads it                                      1
  store_data_overwrite  VMEM store of > 64 bits        -> VALU overwrites one of its data VGPRs                                 1 (2 with an SGPR offset)
  readlane_vgpr         VALU writes a VGPR             -> v_readlane / v_readfirstlane reads it (vsrc0)                          1 (dst-forwarding rule, gfx940+)
  setreg_getreg         s_setreg                       -> s_getreg of the same register                                         2
(A branch on vccz / execz right behind the VALU that wrote VCC / EXEC is interlocked by the hardware on gfx9 and later; the 5-wait-state rule of
the ISA guide is about VALU instructions that read VCCZ / EXECZ as DATA, which the compiler never emits: not scanned.)
Wait states between two instructions = the instructions issued in between, an `s_nop N` counting N + 1.  The scan is linear within a
function; a label resets nothing (the look-back simply continues: fall-through order), branches are not followed — a hazard across a taken
branch would need the CFG, and is reported as unscanned when a consumer has no producer in range.

usage: python tools/hazard_scan.py listing.s [other.s]   — with two listings the per-pattern minimum distances and violation counts are
printed side by side, then every DS instruction count with a folded immediate offset (a second workgroup on a CU has a non-zero LDS base)."""
import collections
import re
import sys

SGPR = re.compile(r'\bs(\d+)\b|\bs\[(\d+):(\d+)\]|\b(vcc|exec|m0)(_lo|_hi)?\b')
VGPR = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
TRANS = ('v_exp_', 'v_log_', 'v_rcp_', 'v_rsq_', 'v_sqrt_', 'v_sin_', 'v_cos_')
REQ = {'valu_sgpr_vmem': 5, 'valu_sgpr_lane': 4, 'valu_vcc_divfmas': 4, 'valu_exec_dpp': 5, 'valu_vgpr_dpp': 2, 'salu_m0_lds': 1,
       'trans_use': 1, 'store_data_overwrite': 1, 'readlane_vgpr': 1, 'setreg_getreg': 2}


def regs(text, rx):
    out = set()
    for m in rx.finditer(text):
        g = m.groups()
        if rx is SGPR:
            if g[0] is not None:
                out.add('s%d' % int(g[0]))
            elif g[1] is not None:
                out.update('s%d' % i for i in range(int(g[1]), int(g[2]) + 1))
            else:
                out.add(g[3])
        else:
            if g[0] is not None:
                out.add('v%d' % int(g[0]))
            else:
                out.update('v%d' % i for i in range(int(g[1]), int(g[2]) + 1))
    return out


def split_ops(line):
    parts = line.split(None, 1)
    mnem = parts[0]
    ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
    # re-join register ranges split by the comma inside brackets never happens: v[0:1] has no comma
    return mnem, ops


def is_valu(m):
    return m.startswith('v_')


def is_vmem(m):
    return m.startswith(('global_', 'flat_', 'scratch_', 'buffer_', 'tbuffer_', 'image_'))


def scan(path):
    txt = open(path).read()
    stats = {k: {'min': None, 'viol': 0, 'seen': 0, 'examples': []} for k in REQ}
    ds_off = collections.Counter()
    for m in re.finditer(r'^([A-Za-z_][\w$.]*):[^\n]*\n(.*?)(?=^\s*\.(?:Lfunc_end|section|size)\b)', txt, re.S | re.M):
        fn, body = m.group(1), m.group(2)
        if fn.startswith('.L'):
            continue
        ins = []
        for line in body.split('\n'):
            line = line.split(';')[0].strip()
            if not line or line[0] == '.' or line.endswith(':'):
                continue
            ins.append(line)
        pos = 0                                    # wait-state clock
        last = {}                                  # (kind, reg) -> clock of the producing instruction
        store_data = {}                            # vgpr -> (clock, need)
        for line in ins:
            mnem, ops = split_ops(line)
            if mnem == 's_nop':
                pos += int(ops[0], 0) + 1
                continue
            pos += 1

            def note(kind, prod_clock, detail):
                d = pos - prod_clock - 1            # wait states in between
                st = stats[kind]
                st['seen'] += 1
                if st['min'] is None or d < st['min']:
                    st['min'] = d
                need = REQ[kind]
                if d < need:
                    st['viol'] += 1
                    if len(st['examples']) < 4:
                        st['examples'].append('%s: %s (distance %d < %d) %s' % (fn[:60], line[:70], d, need, detail))
            dst = ops[0] if ops else ''
            srcs = ', '.join(ops[1:]) if len(ops) > 1 else ''
            if mnem.startswith('ds_') and 'offset:' in line:
                ds_off[fn] += 1
            # ---------------- consumers
            if is_vmem(mnem):
                for r in regs(line, SGPR):
                    if ('valu_s', r) in last:
                        note('valu_sgpr_vmem', last[('valu_s', r)], r)
            if mnem.startswith(('v_readlane', 'v_writelane')):
                sel = ops[-1] if ops else ''
                for r in regs(sel, SGPR):
                    if ('valu_s', r) in last:
                        note('valu_sgpr_lane', last[('valu_s', r)], r)
            if mnem.startswith('v_div_fmas') and ('valu_s', 'vcc') in last:
                note('valu_vcc_divfmas', last[('valu_s', 'vcc')], 'vcc')
            if is_valu(mnem) and ('dpp' in line or 'quad_perm' in line or 'row_' in line):
                if ('valu_s', 'exec') in last:
                    note('valu_exec_dpp', last[('valu_s', 'exec')], 'exec')
                for r in regs(srcs, VGPR):
                    if ('valu_v', r) in last:
                        note('valu_vgpr_dpp', last[('valu_v', r)], r)
            if (mnem.startswith('ds_') and ('gds' in line or 'addtid' in mnem)) or mnem == 's_sendmsg':
                if ('salu_s', 'm0') in last:
                    note('salu_m0_lds', last[('salu_s', 'm0')], 'm0')
            if is_valu(mnem) and not mnem.startswith(TRANS):
                for r in regs(srcs, VGPR):
                    if ('trans', r) in last:
                        note('trans_use', last[('trans', r)], r)
            if mnem.startswith(('v_readlane', 'v_readfirstlane')) and len(ops) > 1:
                for r in regs(ops[1], VGPR):
                    if ('valu_v', r) in last:
                        note('readlane_vgpr', last[('valu_v', r)], r)
            if mnem.startswith('s_getreg') and ('setreg', 'any') in last:
                note('setreg_getreg', last[('setreg', 'any')], '')
            if is_valu(mnem):
                for r in regs(dst, VGPR):
                    if r in store_data:
                        clk, need = store_data[r]
                        d = pos - clk - 1
                        st = stats['store_data_overwrite']
                        st['seen'] += 1
                        if st['min'] is None or d < st['min']:
                            st['min'] = d
                        if d < need:
                            st['viol'] += 1
                            if len(st['examples']) < 4:
                                st['examples'].append('%s: %s (distance %d < %d)' % (fn[:60], line[:70], d, need))
            # ---------------- producers
            if is_valu(mnem):
                # destination(s): vdst and, for VOP3 carry / compare forms, an SGPR pair or vcc as the first or second operand
                for r in regs(dst, VGPR):
                    last[('valu_v', r)] = pos
                    last.pop(('trans', r), None)
                    if mnem.startswith(TRANS):
                        last[('trans', r)] = pos
                sdst = set()
                if mnem.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
                    sdst |= regs(dst, SGPR)
                    if mnem.startswith('v_cmp') and mnem.endswith('_e32'):
                        sdst.add('vcc')
                    if mnem.startswith('v_cmpx'):
                        sdst.add('exec')
                if mnem.startswith(('v_mad_u64_u32', 'v_mad_i64_i32', 'v_add_co', 'v_sub_co', 'v_subrev_co', 'v_addc_co', 'v_subb_co', 'v_div_scale')) and len(ops) > 1:
                    sdst |= regs(ops[1], SGPR)
                    if mnem.endswith('_e32'):
                        sdst.add('vcc')
                for r in sdst:
                    last[('valu_s', r)] = pos
                    if r in ('vcc_lo', 'vcc_hi'):
                        last[('valu_s', 'vcc')] = pos
            elif mnem.startswith('s_') and not mnem.startswith(('s_waitcnt', 's_cbranch', 's_branch', 's_barrier', 's_endpgm', 's_setpc', 's_swappc', 's_sleep')):
                for r in regs(dst, SGPR):
                    last[('salu_s', r)] = pos
                    last.pop(('valu_s', r), None)          # a later SALU write supersedes the VALU's
                if mnem.startswith('s_setreg'):
                    last[('setreg', 'any')] = pos
            if is_vmem(mnem) and 'store' in mnem and ('x3' in mnem or 'x4' in mnem):
                data = ops[1] if mnem.startswith(('global_', 'flat_')) and len(ops) > 1 else (ops[1] if mnem.startswith('scratch_') and len(ops) > 1 else (ops[0] if ops else ''))
                need = 2 if re.search(r'\bs\d+\b|\bs\[', ', '.join(ops[2:])) else 1
                for r in regs(data, VGPR):
                    store_data[r] = (pos, need)
            for r in list(store_data):
                if pos - store_data[r][0] > 4:
                    del store_data[r]
    return stats, ds_off


def main():
    paths = sys.argv[1:]
    res = [scan(p) for p in paths]
    print('%-22s %s' % ('pattern (need)', '   '.join('%-34s' % p[-34:] for p in paths)))
    for k in REQ:
        row = '%-22s' % ('%s (%d)' % (k, REQ[k]))
        for st, _ in res:
            s = st[k]
            row += '   %-34s' % ('seen %6d  min %s  below %d' % (s['seen'], '-' if s['min'] is None else s['min'], s['viol']))
        print(row)
    for (st, _), p in zip(res, paths):
        for k in REQ:
            for e in st[k]['examples']:
                print('  %s %s: %s' % (p[-20:], k, e))
    if len(res) == 2:
        a, b = res[0][1], res[1][1]
        diff = [(f, a[f], b[f]) for f in sorted(set(a) | set(b)) if a[f] != b[f]]
        print('DS instructions with a folded immediate offset: %d vs %d in total; functions that differ: %d' % (sum(a.values()), sum(b.values()), len(diff)))
        for f, x, y in diff[:12]:
            print('   %-70s %d vs %d' % (f[:70], x, y))


if __name__ == '__main__':
    main()
