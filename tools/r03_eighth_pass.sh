set -o pipefail
O=gpurun_out/r03h; mkdir -p $O
timeout -k 10 300 python tools/clock_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/clock_probe.txt
timeout -k 10 300 python tools/shim_latency.py 2>&1 | grep -v amdgpu.ids | tee $O/shim_latency.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
