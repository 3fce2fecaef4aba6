#!/bin/bash
# One box's record for the question "why does g1_mul_kernel lose 6 % on some boxes at an equal or higher clock": the kernel's time without a
# profiler (tools/ab_bench.py --legs g1, 2^20) and, from four counter passes over tools/prof_driver3.py g1 (2^18 = one launch of two machine
# rounds): wait fractions, the average latency of a vector-memory read (SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM_RD: the 176-byte table records),
# L2 hit rate, HBM-side fetch bytes (doubled per the gfx950 rule) and the clock held (GRBM_GUI_ACTIVE / 8 / duration).
#   usage (GPU box): bash tools/box_g1_counters.sh <outdir under the repository>
OUT=${1:-gpurun_out/box_g1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd "$ROOT"
{ hostname; rocm-smi --showuniqueid --showclocks --showpower --showtemp 2>/dev/null | grep -v "^=\|^$" | head -40; } > "$OUT/box.txt" 2>&1
timeout -k 10 300 python3 tools/ab_bench.py --legs g1 --rounds 2 --reps 5 default > "$OUT/g1_time.txt" 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "FETCH_SIZE" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$ROOT/$OUT/$tag" -o p -- python3 "$ROOT/tools/prof_driver3.py" g1 > "$ROOT/$OUT/$tag.log" 2>&1
  rc=$?
  echo "pass $tag rc=$rc"
  [ $rc -ne 0 ] && exit $rc
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for path in glob.glob("%s/*/*counter_collection.csv" % out) + glob.glob("%s/*/*/*counter_collection.csv" % out):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if "g1_mul_kernel" not in r["Kernel_Name"]:
            continue
        k = r["Dispatch_Id"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        agg[k]["dur_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if not agg:
        continue
    v = max(agg.values(), key=lambda d: d["dur_ms"])
    tag = path.split("/")[-3] if path.count("/") > out.count("/") + 2 else path.split("/")[-2]
    for k, x in v.items():
        res[k if k != "dur_ms" else "dur_ms_" + tag] = x
d = dict(res)
if d.get("SQ_WAVE_CYCLES"):
    d["wait_any_frac"] = d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]
    d["wait_inst_any_frac"] = d.get("SQ_WAIT_INST_ANY", 0) / d["SQ_WAVE_CYCLES"]
if d.get("SQ_INSTS_VMEM_RD"):
    d["vmem_read_latency_cycles"] = d.get("SQ_INST_LEVEL_VMEM", 0) / d["SQ_INSTS_VMEM_RD"]
if d.get("TCC_HIT_sum") is not None and (d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0)) > 0:
    d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
if "FETCH_SIZE" in d:
    d["hbm_fetch_GB"] = 2 * d["FETCH_SIZE"] * 1024 / 1e9
for k in list(d):
    if k.startswith("dur_ms_GRBM") and "GRBM_GUI_ACTIVE" in d:
        d["clock_GHz"] = d["GRBM_GUI_ACTIVE"] / 8 / (d[k] * 1e-3) / 1e9
json.dump(d, open("%s/summary.json" % out, "w"), indent=1)
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if not k.startswith("SQ_") and not k.startswith("TCC_") and k not in ("FETCH_SIZE", "GRBM_GUI_ACTIVE")}))
PY
grep -E "^g1 " "$OUT/g1_time.txt"
