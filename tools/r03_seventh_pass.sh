set -o pipefail
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 300 python tools/clock_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/clock_probe.txt
