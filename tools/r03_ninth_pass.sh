set -o pipefail
O=gpurun_out/r03i; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_full_batch.py tests/test_gpu_pairing.py tests/test_gpu_g1.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03i/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_vs_round2_peak"])
for k in ("pairing","g2_mul","miller","fexp","msm","bbs_plus"):
    if k in d: print(k, d[k]["value"], d[k]["ms_per_step"], d[k].get("roofline",{}).get("frac"), d[k].get("roofline",{}).get("frac_vs_round2_peak"), d[k].get("roofline",{}).get("avg_launch_ms"))
print(d["msm"]["roofline_whole_step"]["frac_vs_round2_peak"])
PY
