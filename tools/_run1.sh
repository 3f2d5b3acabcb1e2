timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_ae_gpu.py -x -q 2>&1 | tail -3
for k in stream coop; do echo KERNEL $k; P2P_SMALL_KERNEL=$k timeout 300 python tools/time_small.py resnet50 50 1,3,8 2>&1 | grep resnet; done
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $G/prof_small1
rocprofv3 --kernel-trace -d $G/prof_small1 -o t -- python $GRAFT_REPO_ROOT/tools/time_small.py resnet50 3 1 > $G/small1.log 2>&1
