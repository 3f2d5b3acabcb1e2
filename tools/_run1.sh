cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $G/prof_call
rocprofv3 --kernel-trace -d $G/prof_call -o t -- python $GRAFT_REPO_ROOT/tools/single_det.py 10 > $G/call.log 2>&1
