# quick check after a kernel change: the G1 / compressed-input / full-batch tests, then the A/B timing tool on the default library
mkdir -p gpurun_out/r03l
timeout -k 10 900 python -m pytest tests/test_gpu_g1.py tests/test_gpu_compressed_in.py tests/test_gpu_full_batch.py tests/test_gpu_dropin.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r03l/pytest.log; echo "pytest rc=$?"; cat gpurun_out/r03l/pytest.log
timeout -k 10 200 python tools/g2_mul_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03l/bench.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sampled-parity > gpurun_out/r03l/bench.json 2> gpurun_out/r03l/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03l/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
PY
