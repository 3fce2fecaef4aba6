# A/B harness: run bench.py against alternative builds of the same ABI (C12381_LIB) in one GPU session.
# usage: bash tools/ab_variants.sh "<variant names under crypto12381_amd/lib/exp/>" [extra bench args]
mkdir -p gpurun_out
VARIANTS=${1:-"base"}
shift
for v in $VARIANTS; do
  if [ $v = base ]; then unset C12381_LIB; else export C12381_LIB=$GRAFT_REPO_ROOT/crypto12381_amd/lib/exp/lib$v.so; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-msm --sampled-parity "$@" > gpurun_out/ab_$v.json 2>gpurun_out/ab_$v.err || { tail -3 gpurun_out/ab_$v.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json"))
p=d.get("pairing"); b=d.get("bbs_plus"); print("$v", "g1 %.3e /s (kernel %.2f ms)" % (d["value"], d["roofline"]["avg_launch_ms"]), ("pair %.3e /s (kernel %.2f ms)" % (p["value"], p["roofline"]["avg_launch_ms"])) if p else "", ("bbs %.2f ms" % b["ms_per_step"]) if b else "")
PY
done
