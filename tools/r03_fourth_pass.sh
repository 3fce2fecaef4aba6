set -o pipefail
O=gpurun_out/r03d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_compressed_in.py tests/test_gpu_g1.py tests/test_gpu_full_batch.py -m gpu -x -q 2>&1 | tail -25 > $O/pytest_gpu.log; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log
timeout -k 10 300 python tools/g2_mul_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_default.txt
