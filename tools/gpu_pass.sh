#!/bin/bash
# One parametrised pass on the GPU box (replaces the per-experiment r03_*_pass.sh scripts).  Stages run in the order given and the pass stops at
# the first stage that fails or times out (no GPU step is started after a killed one).  Outputs under gpurun_out/<tag>/.
#   bash tools/gpu_pass.sh <tag> <stage> [<stage> ...]
# stages:
#   tests[:<pytest args, comma separated>]   pytest -m gpu (default: the whole suite)
#   quick                                    the G1 / G2 / pairing / full-batch / drop-in tests only
#   bench[:<bench args, comma separated>]    python bench.py (default --steps 20 --warmup 5)
#   ab:<lib>[,<lib>...]                      tools/ab_bench.py default <libs>   (names under crypto12381_amd/lib/exp/ or paths)
#   abl:<legs>:<lib>[,<lib>...]              the same with --legs
#   prof                                     rocprofv3 --kernel-trace --stats over bench.py --steps 5 --warmup 2
#   pmc                                      tools/pmc_r03.sh counter passes (traffic, issue counters)
#   clock                                    tools/clock_probe.py
#   2rank                                    bench.py --gpus 2 started plainly, gloo on the one GPU
#   soak / psoak:<minutes>                   tools/soak.py / tools/parity_soak.py
#   wstats[:<n>]                             tools/queue_wave_stats.py (experiments build): where the queue kernels' wavefronts spend a launch
#   boxg1                                    tools/box_g1_counters.sh: this box's G1 time, wait fractions, gather latency, L2 hit rate, clock
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
libpath() { case "$1" in default) echo default;; */*) echo "$1";; *) echo crypto12381_amd/lib/exp/lib$1.so;; esac; }
for st in "$@"; do
  name=${st%%:*}; arg=""; [ "$st" != "$name" ] && arg=${st#*:}
  echo "== stage $st"
  case $name in
    tests) timeout -k 10 1100 python -m pytest tests -m gpu -q ${arg//,/ } 2>&1 | tail -25 > $O/pytest_gpu.log; rc=$?; tail -4 $O/pytest_gpu.log;;
    quick) timeout -k 10 900 python -m pytest tests/test_gpu_g1.py tests/test_gpu_pairing.py tests/test_gpu_full_batch.py tests/test_gpu_dropin.py tests/test_gpu_bbs.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest_quick.log; rc=$?; tail -3 $O/pytest_quick.log;;
    bench) a=${arg//,/ }; [ -z "$a" ] && a="--steps 20 --warmup 5"
           timeout -k 10 900 python bench.py $a > $O/bench.json 2> $O/bench.err; rc=$?; tail -2 $O/bench.err; python tools/bench_summary.py $O/bench.json;;
    ab)    libs=""; for l in ${arg//,/ }; do libs="$libs $(libpath $l)"; done
           timeout -k 10 1100 python tools/ab_bench.py default $libs 2>&1 | grep -v amdgpu.ids | tee $O/ab_$(echo $arg | tr ',/' '__').txt; rc=$?;;
    abl)   legs=${arg%%:*}; rest=${arg#*:}; libs=""; for l in ${rest//,/ }; do libs="$libs $(libpath $l)"; done
           timeout -k 10 1100 python tools/ab_bench.py --legs $legs default $libs 2>&1 | grep -v amdgpu.ids | tee $O/ab_$(echo $rest | tr ',/' '__').txt; rc=$?;;
    prof)  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --sampled-parity --no-streamed > $O/bench_under_rocprof.json 2> $O/rocprof.err); rc=$?;;
    pmc)   bash tools/pmc_r03.sh gpurun_out/$TAG/pmc > $O/pmc.txt 2>&1; rc=$?; tail -8 $O/pmc.txt;;
    clock) timeout -k 10 300 python tools/clock_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/clock_probe.txt; rc=$?;;
    2rank) C12381_BENCH_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; rc=$?; python tools/bench_summary.py $O/bench_2rank_gloo.json;;
    soak)  timeout -k 10 400 python3 tools/soak.py > $O/soak.log 2>&1; rc=$?; tail -2 $O/soak.log;;
    psoak) timeout -k 10 1000 python3 tools/parity_soak.py --minutes ${arg:-2.5} > $O/parity_soak.log 2>&1; rc=$?; tail -1 $O/parity_soak.log;;
    wstats) C12381_LIB=crypto12381_amd/lib/libc12381_hip_exp.so C12381_PAIR_STAMPS=/tmp/c12381_stamps_$$.bin timeout -k 10 400 python3 tools/queue_wave_stats.py ${arg} 2>&1 | grep -v amdgpu.ids | tee $O/queue_wave_stats.txt; rc=$?;;
    boxg1) bash tools/box_g1_counters.sh gpurun_out/$TAG/box_g1 2>&1 | tee $O/box_g1.txt; rc=$?;;
    *) echo "unknown stage $name"; rc=2;;
  esac
  echo "== stage $st rc=$rc"
  [ $rc -ne 0 ] && exit $rc
done
exit 0
