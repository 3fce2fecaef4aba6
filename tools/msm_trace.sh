# kernel trace of a few 2^22-term bucket products (tools/msm_only.py): where the time between the kernels goes
mkdir -p gpurun_out/r03p
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r03p/trace -o t -- python3 $ROOT/tools/msm_only.py > $ROOT/gpurun_out/r03p/traced.txt 2>&1; echo "rocprof rc=$?"; grep msm $ROOT/gpurun_out/r03p/traced.txt
