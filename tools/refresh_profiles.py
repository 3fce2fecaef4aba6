#!/usr/bin/env python3
"""Copy the outputs of tools/final_pass.sh (gpurun_out/final/) into profiles/ under the round's names and refresh
profiles/traffic.json from the counter passes.  usage: python3 tools/refresh_profiles.py [round-prefix, default r03]"""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
R = sys.argv[1] if len(sys.argv) > 1 else "r03"
F = "gpurun_out/final"


def short(n):
    n = re.sub(r"^void ", "", n.split("(")[0])
    if "rocprim" in n:
        m = re.search(r"rocprim::(?:detail::)?(\w+)", n)
        n = "rocprim::" + (m.group(1) if m else "kernel")
    if n.startswith("at::") or "at::native" in n:
        n = "torch kernel"
    return n


rows = list(csv.DictReader(open(F + "/prof/p_kernel_stats.csv")))
with open("profiles/%s_kernel_stats_bench.csv" % R, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --sampled-parity  (MI355X, final build of the round, "
            "tools/final_pass.sh; rocPRIM / torch kernel names shortened; bench line of the same run: profiles/%s_bench_under_rocprof.json)\n" % R)
    f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs\n")
    for r in rows:
        f.write("%s,%s,%s,%s,%s,%s\n" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
for src, dst in (("bench_under_rocprof.json", "bench_under_rocprof.json"), ("bench.json", "bench_final.json"), ("pmc/summary.json", "pmc_summary.json"),
                 ("soak.log", "soak.log"), ("pytest_gpu.log", "pytest_gpu_final.log"), ("bench_2rank_gloo.json", "bench_2rank_gloo.json"),
                 ("clock_probe.txt", "clock_probe.txt"), ("pmc.txt", "pmc_passes.txt")):
    if os.path.exists(F + "/" + src):
        shutil.copy(F + "/" + src, "profiles/%s_%s" % (R, dst))
s = json.load(open(F + "/pmc/summary.json"))
t = json.load(open("profiles/traffic.json"))
g = s["g1_mul_kernel"]
t["FETCH_SIZE_KB"], t["WRITE_SIZE_KB"] = g["FETCH_SIZE"], g["WRITE_SIZE"]
t["g1_mul_kernel_hbm_bytes_per_launch"] = g["FETCH_SIZE"] * 2048 + g["WRITE_SIZE"] * 1024
t["units_per_launch"] = 262144          # tools/prof_driver3.py: 2^18 G1 elements, ONE launch since launches span eight rounds
for key, kern in (("pair_kernel", "pair3_queue_kernel"), ("pair3_prod_fixed_queue_kernel", "pair3_prod_fixed_queue_kernel"),
                  ("msm_bucket_kernel", "msm_bucket_kernel"), ("g2_mul2_kernel", "g2_mul2_kernel")):
    if kern in s and "FETCH_SIZE" in s[kern]:
        t[key]["FETCH_SIZE_KB"], t[key]["WRITE_SIZE_KB"] = s[kern]["FETCH_SIZE"], s[kern]["WRITE_SIZE"]
        t[key]["hbm_bytes_per_launch"] = s[kern]["FETCH_SIZE"] * 2048 + s[kern]["WRITE_SIZE"] * 1024
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
# profiles/issue.json: what bench.py's `roofline.issue` object is computed from — VALU instructions per launch (SQ_INSTS_VALU pass) and the
# clock the chip holds inside each kernel (tools/clock_probe.py); the cycles per instruction come from csrc/microbench/issue_mix.hip
clk = {}
for line in open("profiles/%s_clock_probe.txt" % R):
    m = re.match(r"(\w+) .*median ([\d.]+) GHz", line)
    if m:
        clk[m.group(1)] = float(m.group(2))
issue = {"source": "SQ_INSTS_VALU: rocprofv3 --pmc pass of tools/pmc_r03.sh (profiles/%s_pmc_summary.json); clock_GHz: median of tools/clock_probe.py "
                   "(profiles/%s_clock_probe.txt; the BBS+ kernel takes the pairing kernel's); cycles_per_valu_inst: csrc/microbench/issue_mix.hip, "
                   "profiles/r03_issue_mix.txt — a SIMD issues one vector instruction per 4.06 cycles from its OLDEST wavefront whatever the "
                   "instruction is; a second wavefront only fills the older one's stalls" % (R, R),
         "cycles_per_valu_inst": 4.06, "simds": 1024, "kernels": {}}
for kern, units, ck in (("g1_mul_kernel", 262144, "g1_mul"), ("g2_mul2_kernel", 131072, "g2_mul"), ("pair3_queue_kernel", 65536, "pairing"),
                        ("miller3_queue_kernel", 65536, "miller"), ("fexp3_queue_kernel", 65536, "fexp"), ("msm_bucket_kernel", 4194304, "msm"),
                        ("pair3_prod_fixed_queue_kernel", 262144, "pairing")):
    if kern in s and "SQ_INSTS_VALU" in s[kern] and ck in clk:
        issue["kernels"][kern] = {"units_per_launch": units, "valu_insts_per_launch": s[kern]["SQ_INSTS_VALU"], "clock_GHz": clk[ck],
                                  "wait_any_frac": s[kern].get("wait_any_frac")}
json.dump(issue, open("profiles/issue.json", "w"), indent=1)
for k in ("g1_mul_kernel", "g2_mul2_kernel", "pair3_queue_kernel", "pair3_prod_fixed_queue_kernel", "msm_bucket_kernel", "miller3_queue_kernel", "fexp3_queue_kernel"):
    if k in s:
        v = s[k]
        print("%-32s HBM-side %.2f GB  VALU %.3e  wait %.3f" % (k, v.get("hbm_bytes_per_launch", 0) / 1e9, v.get("SQ_INSTS_VALU", 0), v.get("wait_any_frac", 0)))
for r in rows[:8]:
    print("%-50s %5s calls  avg %.3f ms" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e6))
d = json.load(open(F + "/bench.json"))
print("bench: g1 %.3e /s %.2f ms (frac %.3f / %.3f) | pairing %.3e /s %.2f ms (%.3f / %.3f)" % (
    d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_vs_round2_peak"], d["pairing"]["value"], d["pairing"]["ms_per_step"],
    d["pairing"]["roofline"]["frac"], d["pairing"]["roofline"]["frac_vs_round2_peak"]))
for k in ("g2_mul", "miller", "fexp", "msm", "bbs_plus"):
    print("   %-9s %.3e /s  %.2f ms  frac %.3f / %.3f  kernel %.2f ms" % (k, d[k]["value"], d[k]["ms_per_step"], d[k]["roofline"]["frac"], d[k]["roofline"]["frac_vs_round2_peak"],
                                                                          d[k]["roofline"]["avg_launch_ms"]))
