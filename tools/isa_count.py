#!/usr/bin/env python3
"""Static instruction classes of the functions of a hipcc -S listing: multiply-adds, other VALU by mnemonic, SALU, waits, memory.
usage: python tools/isa_count.py listing.s [function-name-regex]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for m in re.finditer(r'^([A-Za-z_][\w$.]*):[^\n]*\n(.*?)(?=^\s*\.(?:Lfunc_end|section|size)\b)', txt, re.S | re.M):
    fn, body = m.group(1), m.group(2)
    if fn.startswith('.L') or (pat and not pat.search(fn)):
        continue
    c = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        if not line or line[0] in ';.' or line.endswith(':'):
            continue
        c[line.split()[0]] += 1
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    if valu < 50:
        continue
    mad = sum(v for k, v in c.items() if k.startswith('v_mad_i64') or k.startswith('v_mad_u64'))
    other = {k: v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mad_i64') and not k.startswith('v_mad_u64')}
    print("%s: VALU %d  mad64 %d  other %d (%.3f per mad)  SALU %d  nop %d  waitcnt %d  scratch %d  ds %d  global/flat %d" % (
        fn, valu, mad, valu - mad, (valu - mad) / max(mad, 1), sum(v for k, v in c.items() if k.startswith('s_') and k not in ('s_nop', 's_waitcnt')),
        c['s_nop'], c['s_waitcnt'], sum(v for k, v in c.items() if k.startswith('scratch_')), sum(v for k, v in c.items() if k.startswith('ds_')),
        sum(v for k, v in c.items() if k.startswith(('global_', 'flat_', 'buffer_')))))
    print("   " + "  ".join("%s %d" % kv for kv in sorted(other.items(), key=lambda kv: -kv[1])[:14]))
