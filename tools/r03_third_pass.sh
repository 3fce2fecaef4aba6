# round 3, third GPU pass: A/B of the linear accumulator chains (fence) and of the inlined two-lane G2 loop, then the whole GPU suite
set -o pipefail
O=gpurun_out/r03c; mkdir -p $O
for v in default nofence g2noinl default nofence; do
  if [ $v = default ]; then unset C12381_LIB; else export C12381_LIB=$PWD/crypto12381_amd/lib/exp/lib$v.so; fi
  echo "== $v" >> $O/ab_fence.txt
  timeout -k 10 300 python tools/g2_mul_bench.py 2>&1 | grep -v amdgpu.ids >> $O/ab_fence.txt || exit 1
done
unset C12381_LIB
cat $O/ab_fence.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
