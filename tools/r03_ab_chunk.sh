# A/B: scalar-multiplication launches of one machine round (default) against eight rounds per launch (libchunk8.so)
mkdir -p gpurun_out/r03k
for rep in 1 2; do
  for v in default chunk8; do
    echo "== $v"
    if [ $v = default ]; then timeout -k 10 200 python tools/g2_mul_bench.py 2>&1 | grep -v amdgpu.ids
    else C12381_LIB=crypto12381_amd/lib/exp/lib$v.so timeout -k 10 200 python tools/g2_mul_bench.py 2>&1 | grep -v amdgpu.ids; fi
  done
done > gpurun_out/r03k/ab_chunk.txt 2>&1
cat gpurun_out/r03k/ab_chunk.txt
