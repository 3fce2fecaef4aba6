#!/usr/bin/env python3
"""Print the numbers of a bench.py JSON line that an A/B or a round summary needs (one row per leg) and the line's length."""
import json
import sys

raw = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")]
if not raw:
    sys.exit("no JSON line in %s" % sys.argv[1])
d = json.loads(raw[-1])
print("line: %d characters, last key %s, n_gpus %s, rccl_ranks %s" % (len(raw[-1]), list(d)[-1], d.get("n_gpus"), d.get("rccl_ranks")))


def row(name, o):
    r = o.get("roofline", {})
    i = r.get("issue", {})
    print("%-14s value %.4g %s  %.3f ms/step  kernel %s %.3f ms  frac %.3f  clock %s  issue/launch %s  cpu %s" % (
        name, o["value"], o["unit"], o["ms_per_step"], r.get("kernel", "-"), r.get("avg_launch_ms", 0), r.get("frac", 0), i.get("clock_GHz_in_run"),
        ("%.3f" % i["issue_over_launch"]) if "issue_over_launch" in i else "-", ("%.4g" % o["cpu_baseline"]["value"]) if "cpu_baseline" in o else "-"))


row("g1 (headline)", d)
for k in ("pairing", "g2_mul", "miller", "fexp", "msm", "bbs_plus", "bbs_plus_wire", "msm_sharded", "bbs_plus_sharded"):
    if k in d:
        row(k, d[k])
if "msm" in d:
    print("msm whole-step frac %.3f" % d["msm"]["roofline_whole_step"]["frac"])
