#!/bin/bash
# Build an alternative libc12381_hip.so with extra compile flags into crypto12381_amd/lib/exp/lib<name>.so (A/B runs: C12381_LIB).
# usage: bash tools/build_variant.sh <name> <extra flags...>      e.g.  bash tools/build_variant.sh dflt BASEFLAGS=... (the compile-time A/B knobs of
# rounds 2-4 are gone from the sources with their losing arms: a variant is a working-tree edit built under another name, or other compiler flags)
# (add -DC12381_EXPERIMENTS for a variant that also reads the C12381_* tuning variables; BASEFLAGS="..." in the environment replaces the
# product's base flags, e.g. to drop the max-ilp scheduling strategy)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/crypto12381_amd/lib/exp; OBJ=$OUT/obj_$NAME
mkdir -p $OBJ
FLAGS=${BASEFLAGS:-"-O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-optimize-sibling-calls -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -opt-disable=reassociate"}
for f in "$@"; do case "$f" in *amdgpu-use-amdgpu-trackers*) echo "refused: $f (docs/lab_notes.md 5b)"; exit 2;; esac; done
pids=""
for u in c12381_hip k_g1 k_g2gt k_g2h k_pair3 k_hash_zp k_fixed; do
  UF=$FLAGS
  # the product builds k_g1.hip with the default scheduling strategy (crypto12381_amd/build.py: DEFAULT_SCHED_UNITS)
  if [ $u = k_g1 ] && [ -z "$BASEFLAGS" ]; then UF=${FLAGS/-mllvm -amdgpu-sched-strategy=max-ilp/}; fi
  /opt/rocm/bin/hipcc $UF "$@" -c -o $OBJ/$u.o $ROOT/crypto12381_amd/csrc/$u.hip 2> $OBJ/$u.err &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib$NAME.so $OBJ/*.o
rm -rf $OBJ
ls -la $OUT/lib$NAME.so
