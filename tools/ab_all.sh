# A/B harness over ALL configs: run bench.py --all-configs against alternative builds of the same ABI (C12381_LIB).
# usage: bash tools/ab_all.sh "<variant names under crypto12381_amd/lib/exp/>"
mkdir -p gpurun_out
for v in ${1:-base}; do
  if [ $v = base ]; then unset C12381_LIB; else export C12381_LIB=$GRAFT_REPO_ROOT/crypto12381_amd/lib/exp/lib$v.so; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --all-configs > gpurun_out/aball_$v.json 2>gpurun_out/aball_$v.err || { tail -3 gpurun_out/aball_$v.err; continue; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/aball_$v.json").read().strip().split("\n")[-1])
p=d.get("pairing"); x=d["extra_configs"]
print("%-14s" % "$v", "g1 %.2f ms" % d["roofline"]["avg_launch_ms"], "pair %.2f ms" % p["roofline"]["avg_launch_ms"], " ".join("%s %.2f" % (k, v.get("ms_per_batch", v.get("ms_per_msm", 0))) for k, v in x.items()))
PY
done
