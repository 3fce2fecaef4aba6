#!/bin/bash
# HBM-side traffic of the two headline kernels: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md HBM
# section) over tools/prof_driver.py (one 2^18 G1 scalar-mul batch = two launches of 2^17, one 2^16 pairing batch), plus a pass of
# SQ issue / wait counters.  Usage (GPU box): bash tools/pmc_traffic.sh <outdir>
OUT=${1:-gpurun_out/pmc_r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$ROOT/$OUT/$tag" -o p -- python3 "$ROOT/tools/prof_driver.py" both > "$ROOT/$OUT/$tag.log" 2>&1
  echo "pass $tag rc=$?"
done
cd "$ROOT"
python3 tools/pmc_summary.py $OUT/FETCH_SIZE/.. 2>/dev/null | head -0
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = collections.defaultdict(dict)
for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES"):
    for path in glob.glob("%s/%s/*counter_collection.csv" % (out, tag)) + glob.glob("%s/%s/*/*counter_collection.csv" % (out, tag)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("c12381::", "")
            if name.startswith("__amd") or "rocprim" in name or "at::" in name:
                continue
            k = (name, r["Dispatch_Id"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            agg[k]["dur_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        # largest-duration dispatch of every kernel
        best = {}
        for (name, d), v in agg.items():
            if name not in best or v["dur_ms"] > best[name]["dur_ms"]:
                best[name] = v
        for name, v in best.items():
            res[name].update(v)
for name in ("g1_mul_kernel", "pair3_queue_kernel", "g1_finish_kernel"):
    if name in res:
        print(name, {k: ("%.6g" % v) for k, v in res[name].items()})
json.dump({k: dict(v) for k, v in res.items()}, open("%s/summary.json" % out, "w"), indent=1)
PY
