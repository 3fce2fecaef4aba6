# round 3, second GPU pass: occupancy A/B (2 vs 3 waves per SIMD; MSM bucket kernel at 4), then the tests touched by the experiments gating
set -o pipefail
O=gpurun_out/r03b; mkdir -p $O
for v in default occ3 occ4msm default occ3; do
  if [ $v = default ]; then unset C12381_LIB; else export C12381_LIB=$PWD/crypto12381_amd/lib/exp/lib$v.so; fi
  echo "== $v" >> $O/ab_occ.txt
  timeout -k 10 300 python tools/g2_mul_bench.py 2>&1 | grep -v amdgpu.ids >> $O/ab_occ.txt || exit 1
done
unset C12381_LIB
cat $O/ab_occ.txt
timeout -k 10 900 python -m pytest tests/test_gpu_variants.py tests/test_gpu_api_contract.py tests/test_gpu_dropin.py tests/test_gpu_distributed.py tests/test_gpu_g1.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
