#!/usr/bin/env python3
"""Copy the outputs of a `tools/gpu_pass.sh <tag> tests bench prof pmc 2rank ...` pass (gpurun_out/<tag>/) into profiles/ under the round's
names and refresh profiles/traffic.json + profiles/issue.json from the counter passes.

    python3 tools/refresh_profiles.py <tag> [round-prefix, default r04]

profiles/issue.json carries the VALU instruction count of every dominant kernel (SQ_INSTS_VALU pass): a property of the BUILD.  The clock
the chip holds inside a kernel is measured by bench.py itself in the run it reports (lib/libc12381_probe.so); the `clock_GHz` kept here
is the profiler-side estimate of the counter pass (GRBM_GUI_ACTIVE / 8 / duration) and is informational only."""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
TAG = sys.argv[1]
R = sys.argv[2] if len(sys.argv) > 2 else "r04"
F = "gpurun_out/" + TAG


def short(n):
    n = re.sub(r"^void ", "", n.split("(")[0])
    if "rocprim" in n:
        m = re.search(r"rocprim::(?:detail::)?(\w+)", n)
        n = "rocprim::" + (m.group(1) if m else "kernel")
    if n.startswith("at::") or "at::native" in n:
        n = "torch kernel"
    return n.replace("c12381::", "")


stats = F + "/prof/p_kernel_stats.csv"
if os.path.exists(stats):
    rows = list(csv.DictReader(open(stats)))
    with open("profiles/%s_kernel_stats_bench.csv" % R, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --sampled-parity  (MI355X, final build of the "
                "round, tools/gpu_pass.sh %s prof; rocPRIM / torch kernel names shortened; bench line of the same run: profiles/%s_bench_under_rocprof.json)\n" % (TAG, R))
        f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs\n")
        for r in rows:
            f.write("%s,%s,%s,%s,%s,%s\n" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
    for r in rows[:10]:
        print("%-50s %5s calls  avg %.3f ms" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e6))
for src, dst in (("bench_under_rocprof.json", "bench_under_rocprof.json"), ("bench.json", "bench_final.json"), ("pmc/summary.json", "pmc_summary.json"),
                 ("soak.log", "soak.log"), ("pytest_gpu.log", "pytest_gpu_final.log"), ("bench_2rank_gloo.json", "bench_2rank_gloo.json"),
                 ("clock_probe.txt", "clock_probe.txt"), ("pmc.txt", "pmc_passes.txt"), ("parity_soak.log", "parity_soak.log")):
    if os.path.exists(F + "/" + src):
        shutil.copy(F + "/" + src, "profiles/%s_%s" % (R, dst))
if os.path.exists(F + "/pmc/summary.json"):
    s = json.load(open(F + "/pmc/summary.json"))
    t = json.load(open("profiles/traffic.json"))
    t["source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_r03.sh over tools/prof_driver3.py: every dominant kernel at its "
                   "BASELINE size), MI355X, round %s build; profiles/%s_pmc_summary.json (longest dispatch of each kernel)" % (R[1:], R))
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
    issue = {"source": "SQ_INSTS_VALU: rocprofv3 --pmc pass of tools/pmc_r03.sh over the round-%s build (profiles/%s_pmc_summary.json); cycles_per_valu_inst: "
                       "csrc/microbench/issue_mix.hip, profiles/r03_issue_mix.txt — a SIMD issues one vector instruction per 4.06 cycles from its OLDEST "
                       "wavefront whatever the instruction is.  clock_GHz here = GRBM_GUI_ACTIVE / 8 / duration of the counter pass (informational): "
                       "bench.py measures the clock inside each kernel in its own run" % (R[1:], R),
             "cycles_per_valu_inst": 4.06, "simds": 1024, "kernels": {}}
    for kern, units in (("g1_mul_kernel", 262144), ("g2_mul2_kernel", 131072), ("pair3_queue_kernel", 65536), ("miller3_queue_kernel", 65536),
                        ("fexp3_queue_kernel", 65536), ("msm_bucket_kernel", 4194304), ("pair3_prod_fixed_queue_kernel", 262144)):
        if kern in s and "SQ_INSTS_VALU" in s[kern]:
            issue["kernels"][kern] = {"units_per_launch": units, "valu_insts_per_launch": s[kern]["SQ_INSTS_VALU"], "clock_GHz": round(s[kern].get("clock_GHz", 2.1), 3),
                                      "wait_any_frac": s[kern].get("wait_any_frac")}
    json.dump(issue, open("profiles/issue.json", "w"), indent=1)
    for k in issue["kernels"]:
        v = s[k]
        print("%-32s HBM-side %.2f GB  VALU %.3e  wait %.3f" % (k, v.get("hbm_bytes_per_launch", 0) / 1e9, v.get("SQ_INSTS_VALU", 0), v.get("wait_any_frac", 0)))
if os.path.exists(F + "/bench.json"):
    os.system("%s tools/bench_summary.py %s/bench.json" % (sys.executable, F))
