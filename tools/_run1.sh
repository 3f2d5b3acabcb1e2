timeout 1500 python -m pytest tests/test_pnp_gpu.py tests/test_est_pose_gpu.py tests/test_golden_gpu.py tests/test_reference_vectors_gpu.py -x -q 2>&1 | tail -3
python tools/single_det.py 100
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $G/prof_call
rocprofv3 --kernel-trace -d $G/prof_call -o t -- python $GRAFT_REPO_ROOT/tools/single_det.py 10 > $G/call.log 2>&1
