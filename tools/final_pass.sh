# Final pass of a round on the GPU box: the whole GPU suite, the default bench line, the same command under rocprofv3 --kernel-trace --stats,
# the counter passes (traffic, clocks, issue counters), the two-rank rehearsal, the clock probe, the soak run.
# Outputs under gpurun_out/final/; tools/refresh_profiles.py copies the summaries into profiles/.
set -o pipefail
mkdir -p gpurun_out/final
timeout -k 10 1000 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/final/pytest_gpu.log; echo "pytest rc=$?"; tail -2 gpurun_out/final/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; echo "bench rc=$?"
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/final/prof -o p -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --sampled-parity > $ROOT/gpurun_out/final/bench_under_rocprof.json 2> $ROOT/gpurun_out/final/rocprof.err; echo "rocprof rc=$?"
cd $ROOT
bash tools/pmc_r03.sh gpurun_out/final/pmc > gpurun_out/final/pmc.txt 2>&1; tail -8 gpurun_out/final/pmc.txt
bash tools/bench_2rank_gloo.sh > gpurun_out/final/bench_2rank_gloo.json 2> gpurun_out/final/bench_2rank_gloo.err; echo "2rank rc=$?"
timeout -k 10 200 python tools/clock_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/final/clock_probe.txt; echo "clock rc=$?"
timeout -k 10 400 python3 tools/soak.py > gpurun_out/final/soak.log 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/final/soak.log
# fresh random inputs, every lane against the compiled reference (CPU-bound: ~70 s per iteration); the long form is run on its own
timeout -k 10 400 python3 tools/parity_soak.py --minutes 2.5 > gpurun_out/final/parity_soak_short.log 2>&1; echo "parity soak rc=$?"; tail -1 gpurun_out/final/parity_soak_short.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/final/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_vs_round2_peak"])
for k in ("pairing","g2_mul","miller","fexp","msm","bbs_plus"):
    if k in d: print(k, d[k]["value"], d[k]["ms_per_step"], d[k].get("roofline",{}).get("frac"), d[k].get("roofline",{}).get("frac_vs_round2_peak"), d[k].get("roofline",{}).get("avg_launch_ms"))
PY
