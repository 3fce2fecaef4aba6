# round 3, first GPU pass: G2 A/B (4- vs 5-bit windows, both on the new arithmetic), VALU calibration, GPU tests, bench
set -o pipefail
O=gpurun_out/r03a; mkdir -p $O
timeout -k 10 120 ./crypto12381_amd/bin/valu_rates $O/valu_rates.txt > $O/valu_rates.log 2>&1; echo "valu_rates rc=$?"
for v in default g2win4 default g2win4; do
  if [ $v = default ]; then unset C12381_LIB; else export C12381_LIB=$PWD/crypto12381_amd/lib/exp/lib$v.so; fi
  echo "== $v" >> $O/ab_g2.txt
  timeout -k 10 200 python tools/g2_mul_bench.py >> $O/ab_g2.txt 2>&1 || exit 1
done
unset C12381_LIB
cat $O/ab_g2.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03a/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k in ("pairing","g2_mul","miller","fexp","msm","bbs_plus"):
    if k in d: print(k, d[k]["value"], d[k]["ms_per_step"], d[k].get("roofline",{}).get("frac"), d[k].get("roofline",{}).get("avg_launch_ms"))
PY
