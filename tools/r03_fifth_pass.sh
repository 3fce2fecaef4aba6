set -o pipefail
O=gpurun_out/r03e; mkdir -p $O
bash tools/pmc_r03.sh $O/pmc > $O/pmc.txt 2>&1; tail -12 $O/pmc.txt
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03e/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_vs_round2_peak"])
for k in ("pairing","g2_mul","miller","fexp","msm","bbs_plus"):
    if k in d: print(k, d[k]["value"], d[k]["ms_per_step"], d[k].get("roofline",{}).get("frac"), d[k].get("roofline",{}).get("avg_launch_ms"), d[k].get("two_full_rounds"))
PY
