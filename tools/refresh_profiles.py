#!/usr/bin/env python3
"""Copy the outputs of tools/final_pass.sh (gpurun_out/final/) into profiles/ under the round's names and refresh
profiles/traffic.json from the counter passes.  usage: python3 tools/refresh_profiles.py [round-prefix, default r02]"""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
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
                 ("soak.log", "soak.log"), ("pytest_gpu.log", "pytest_gpu_final.log"), ("bench_2rank_gloo.json", "bench_2rank_gloo.json")):
    if os.path.exists(F + "/" + src):
        shutil.copy(F + "/" + src, "profiles/%s_%s" % (R, dst))
s = json.load(open(F + "/pmc/summary.json"))
t = json.load(open("profiles/traffic.json"))
g, p = s["g1_mul_kernel"], s["pair3_queue_kernel"]
t["FETCH_SIZE_KB"], t["WRITE_SIZE_KB"] = g["FETCH_SIZE"], g["WRITE_SIZE"]
t["g1_mul_kernel_hbm_bytes_per_launch"] = g["FETCH_SIZE"] * 2048 + g["WRITE_SIZE"] * 1024
t["pair_kernel"]["FETCH_SIZE_KB"], t["pair_kernel"]["WRITE_SIZE_KB"] = p["FETCH_SIZE"], p["WRITE_SIZE"]
t["pair_kernel"]["hbm_bytes_per_launch"] = p["FETCH_SIZE"] * 2048 + p["WRITE_SIZE"] * 1024
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
print("g1 %.2f GB  pairing %.2f GB per launch" % (t["g1_mul_kernel_hbm_bytes_per_launch"] / 1e9, t["pair_kernel"]["hbm_bytes_per_launch"] / 1e9))
print("pairing: SQ_WAIT_ANY/SQ_WAVE_CYCLES %.3f  SQ_INSTS_VALU/SQ_WAVE_CYCLES %.3f  VALU %.3e" % (p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"], p["SQ_INSTS_VALU"] / p["SQ_WAVE_CYCLES"], p["SQ_INSTS_VALU"]))
print("g1: VALU per launch %.3e" % g["SQ_INSTS_VALU"])
for r in rows[:6]:
    print("%-50s %5s calls  avg %.3f ms" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e6))
d = json.load(open(F + "/bench.json"))
print("bench: g1 %.3e /s %.2f ms (frac %.3f) | pairing %.3e /s %.2f ms (frac %.3f) | msm %.2f ms | bbs %.2f ms" % (
    d["value"], d["ms_per_step"], d["valu_roofline"]["frac"], d["pairing"]["value"], d["pairing"]["ms_per_step"], d["pairing"]["valu_roofline"]["frac"],
    d["msm"]["ms_per_step"], d["bbs_plus"]["ms_per_step"]))
