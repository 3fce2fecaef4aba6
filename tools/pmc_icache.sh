#!/bin/bash
# Instruction-cache counters of the two-table pairing kernel (BBS+ leg of bench.py) for up to three library variants:
# usage (GPU box): bash tools/pmc_icache.sh "<variants: base | raw | <name under lib/exp>>" <outdir>
OUT=${2:-gpurun_out/pmc_icache}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
for v in $1; do
  unset C12381_LIB C12381_FQ_RAW
  if [ $v = raw ]; then export C12381_FQ_RAW=1; elif [ $v != base ]; then export C12381_LIB=$ROOT/crypto12381_amd/lib/exp/lib$v.so; fi
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d "$ROOT/$OUT/$v" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-msm --no-pairing --sampled-parity > "$ROOT/$OUT/$v.log" 2>&1
  echo "variant $v rc=$?"
done
cd "$ROOT"
python3 - "$OUT" $1 <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for v in sys.argv[2:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for path in glob.glob("%s/%s/**/*counter_collection.csv" % (out, v), recursive=True):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("c12381::", "")
            if "prod_fixed" not in name and "g1_mul_kernel" not in name:
                continue
            k = (name, r["Dispatch_Id"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            agg[k]["dur_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    best = {}
    for (name, d), c in agg.items():
        if name not in best or c["dur_ms"] > best[name]["dur_ms"]:
            best[name] = c
    for name, c in best.items():
        print("%-6s %-34s %8.2f ms  icache req %.3e hits %.3e misses %.3e (%.1f %%)  ifetch %.3e  wait_inst/wave_cycles %.3f  valu %.3e" % (
            v, name, c["dur_ms"], c["SQC_ICACHE_REQ"], c["SQC_ICACHE_HITS"], c["SQC_ICACHE_MISSES"], 100 * c["SQC_ICACHE_MISSES"] / max(c["SQC_ICACHE_REQ"], 1),
            c["SQ_IFETCH"], c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), c["SQ_INSTS_VALU"]))
PY
