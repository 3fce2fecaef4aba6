# kernel trace of a few 2^22-term bucket products (tools/msm_only.py): where the time between the kernels goes
#   [C12381_LIB=<variant .so>] bash tools/msm_trace.sh <tag> [edge]      -> gpurun_out/<tag>/msm_timeline.txt  ("edge": scalar 1 in lane 1, so the small-scalar bucket is used)
TAG=${1:-msmtrace}
ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/$TAG/trace -o t -- python3 $ROOT/tools/msm_only.py 22 $2 > $ROOT/gpurun_out/$TAG/traced.txt 2>&1; echo "rocprof rc=$?"; grep msm $ROOT/gpurun_out/$TAG/traced.txt
python3 $ROOT/tools/kernel_timeline.py $ROOT/gpurun_out/$TAG/trace > $ROOT/gpurun_out/$TAG/msm_timeline.txt; tail -45 $ROOT/gpurun_out/$TAG/msm_timeline.txt
