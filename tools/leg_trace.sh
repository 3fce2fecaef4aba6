# kernel trace of ONE leg of tools/ab_bench.py (one process, a warm-up call and one timed call) and the timeline of its last call
#   bash tools/leg_trace.sh <tag> <leg: g1|g2|pair|msm|bbs|...> <substring of the first kernel of a call>   -> gpurun_out/<tag>/timeline_<leg>.txt
TAG=$1; LEG=$2; FIRST=$3
ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/$TAG/trace_$LEG -o t -- python3 $ROOT/tools/ab_bench.py --child default --legs $LEG --reps 1 > $ROOT/gpurun_out/$TAG/traced_$LEG.txt 2>&1; echo "rocprof rc=$?"
python3 $ROOT/tools/kernel_timeline.py $ROOT/gpurun_out/$TAG/trace_$LEG $FIRST > $ROOT/gpurun_out/$TAG/timeline_$LEG.txt; grep -v "fillBuffer" $ROOT/gpurun_out/$TAG/timeline_$LEG.txt | tail -60
