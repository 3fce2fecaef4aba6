# generic A/B of the default library against one variant library: bash tools/r03_ab_generic.sh <variant-name>
V=$1
mkdir -p gpurun_out/ab_$V
for rep in 1 2; do
  for v in default $V; do
    echo "== $v"
    if [ $v = default ]; then timeout -k 10 200 python tools/g2_mul_bench.py 2>&1 | grep -v amdgpu.ids
    else C12381_LIB=crypto12381_amd/lib/exp/lib$v.so timeout -k 10 200 python tools/g2_mul_bench.py 2>&1 | grep -v amdgpu.ids; fi
  done
done > gpurun_out/ab_$V/ab.txt 2>&1
cat gpurun_out/ab_$V/ab.txt
