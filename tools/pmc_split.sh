#!/bin/bash
# Traffic past L2 of the pairing's parts: Miller loops alone, final exponentiations alone, queue pairing, plain-grid pairing.
# Usage (GPU box): bash tools/pmc_split.sh <outdir>
OUT=${1:-gpurun_out/pmc_split}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
for q in 1 0; do
export C12381_PAIR_QUEUE=$q
for tag in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $tag --output-format csv -d "$ROOT/$OUT/q$q/$tag" -o p -- python3 "$ROOT/tools/prof_driver.py" split > "$ROOT/$OUT/q${q}_$tag.log" 2>&1
  echo "queue=$q pass $tag rc=$?"
done
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for q in (1, 0):
    res = collections.defaultdict(dict)
    for tag in ("FETCH_SIZE", "WRITE_SIZE"):
        for path in glob.glob("%s/q%d/%s/**/*counter_collection.csv" % (out, q, tag), recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(path)):
                name = r["Kernel_Name"].split("(")[0].replace("c12381::", "")
                k = (name, r["Dispatch_Id"])
                agg[k][tag] += float(r["Counter_Value"])
                agg[k]["dur_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            best = {}
            for (name, d), v in agg.items():
                if name not in best or v["dur_ms"] > best[name]["dur_ms"]:
                    best[name] = v
            for name, v in best.items():
                res[name].update(v)
    print("C12381_PAIR_QUEUE=%d   (GB per launch; FETCH doubled per the gfx950 rule)" % q)
    for name, v in sorted(res.items(), key=lambda kv: -kv[1].get("dur_ms", 0)):
        if v.get("dur_ms", 0) < 0.5:
            continue
        print("  %-34s %8.2f ms  fetch %7.2f GB  write %7.2f GB" % (name, v["dur_ms"], v.get("FETCH_SIZE", 0) * 2048 / 1e9, v.get("WRITE_SIZE", 0) * 1024 / 1e9))
PY
