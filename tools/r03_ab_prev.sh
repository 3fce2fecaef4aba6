# A/B in one session: the working tree's library against crypto12381_amd/lib/exp/libprev.so (the previous commit, same flags)
set -o pipefail
O=gpurun_out/r03k; mkdir -p $O; rm -f $O/ab.txt
for v in default prev default prev; do
  if [ $v = default ]; then unset C12381_LIB; else export C12381_LIB=$PWD/crypto12381_amd/lib/exp/lib$v.so; fi
  echo "== $v" >> $O/ab.txt
  timeout -k 10 300 python tools/g2_mul_bench.py 2>&1 | grep -v amdgpu.ids >> $O/ab.txt || exit 1
done
cat $O/ab.txt
