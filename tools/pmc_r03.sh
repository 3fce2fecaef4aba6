#!/bin/bash
# Round-3 counter passes over tools/prof_driver3.py (every dominant kernel at its BASELINE size): HBM-side traffic (FETCH_SIZE, WRITE_SIZE
# in separate passes, FETCH doubled per the gfx950 rule), the clock the chip holds inside each kernel (GRBM_GUI_ACTIVE / 8 / duration,
# MI355X_MICROARCH.md "DVFS give-back") and the issue counters.  Usage (GPU box): bash tools/pmc_r03.sh <outdir>
OUT=${1:-gpurun_out/pmc_r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$ROOT/$OUT/$tag" -o p -- python3 "$ROOT/tools/prof_driver3.py" all > "$ROOT/$OUT/$tag.log" 2>&1
  echo "pass $tag rc=$?"
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = collections.defaultdict(dict)
for tag in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES"):
    for path in glob.glob("%s/%s/*counter_collection.csv" % (out, tag)) + glob.glob("%s/%s/*/*counter_collection.csv" % (out, tag)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("c12381::", "")
            if name.startswith("__amd") or "rocprim" in name or "at::" in name:
                continue
            k = (name, r["Dispatch_Id"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            agg[k]["dur_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        best = {}
        for (name, d), v in agg.items():          # the longest dispatch of every kernel
            if name not in best or v["dur_ms"] > best[name]["dur_ms"]:
                best[name] = v
        for name, v in best.items():
            for k2, x in v.items():
                res[name][k2 if k2 != "dur_ms" else "dur_ms_%s_pass" % tag] = x
summ = {}
for name, v in res.items():
    d = dict(v)
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = 2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024
    if "GRBM_GUI_ACTIVE" in d:
        d["clock_GHz"] = d["GRBM_GUI_ACTIVE"] / 8 / (d["dur_ms_GRBM_GUI_ACTIVE_pass"] * 1e-3) / 1e9
    if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"]:
        d["wait_any_frac"] = d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]
    summ[name] = d
json.dump(summ, open("%s/summary.json" % out, "w"), indent=1)
for name in ("g1_mul_kernel", "g2_mul2_kernel", "pair3_queue_kernel", "miller3_queue_kernel", "fexp3_queue_kernel", "msm_bucket_kernel", "pair3_prod_fixed_queue_kernel"):
    if name in summ:
        d = summ[name]
        print("%-32s %8.3f ms  clock %.3f GHz  HBM-side %.3f GB  VALU %.4g  wait %.3f" % (
            name, d.get("dur_ms_GRBM_GUI_ACTIVE_pass", d.get("dur_ms_FETCH_SIZE_pass", 0)), d.get("clock_GHz", 0), d.get("hbm_bytes_per_launch", 0) / 1e9,
            d.get("SQ_INSTS_VALU", 0), d.get("wait_any_frac", 0)))
PY
