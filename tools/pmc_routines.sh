#!/bin/bash
# Memory traffic past L2 per call of every three-lane pairing routine: the routine microbenchmark (csrc/microbench/pair_routines.hip,
# one kernel per routine) under separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).  Usage (GPU box): bash tools/pmc_routines.sh <outdir>
OUT=${1:-gpurun_out/pmc_routines}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ITERS=${ITERS:-100}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
for tag in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $tag --output-format csv -d "$ROOT/$OUT/$tag" -o p -- "$ROOT/crypto12381_amd/bin/pair_routines" 2 $ITERS > "$ROOT/$OUT/$tag.log" 2>&1
  echo "pass $tag rc=$?"
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
names = {7: "2x fp_mul", 8: "2x fp2_mul", 5: "fp4_mul_call", 0: "f12t_sqr_h", 1: "miller3_dbl_line", 9: "sqr + dbl_line", 2: "f12t_usqr_h", 3: "f12t_mul_h",
         10: "f12t_mul (private)", 4: "f12t_mul_line_h", 6: "f12t_frob"}
mult = {7: 20, 8: 8, 5: 2, 2: 2, 6: 2}
iters = None
for line in open("%s/FETCH_SIZE.log" % out):
    m = re.search(r"iters\s+(\d+)", line)
    if m and "f12t_sqr_h" in line:
        iters = int(m.group(1))
res = collections.defaultdict(dict)
for tag in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob("%s/%s/**/*counter_collection.csv" % (out, tag), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(path)):
            m = re.search(r"routine_kernel<(\d+)>", r["Kernel_Name"])
            if not m:
                continue
            k = (int(m.group(1)), r["Dispatch_Id"])
            agg[k][tag] += float(r["Counter_Value"])
            agg[k]["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            agg[k]["lanes"] = int(r["Grid_Size"]) if "Grid_Size" in r else 0
        best = {}
        for (kind, d), v in agg.items():
            if kind not in best or v["dur"] > best[kind]["dur"]:
                best[kind] = v
        for kind, v in best.items():
            res[kind][tag] = v[tag]; res[kind]["lanes"] = v["lanes"]
print("routine                 FETCH KB*   WRITE KB   bytes per lane and call (FETCH x2 per the guide's 16-byte rule on gfx950: see fetch_calib)")
for kind, v in sorted(res.items()):
    n = (iters or 100) * mult.get(kind, 1) * max(v.get("lanes", 1), 1)
    f = v.get("FETCH_SIZE", 0) * 1024.0 * 2; w = v.get("WRITE_SIZE", 0) * 1024.0
    print("%-22s %10.0f %10.0f   fetch %8.1f B  write %8.1f B" % (names.get(kind, kind), v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0), f / n, w / n))
PY
