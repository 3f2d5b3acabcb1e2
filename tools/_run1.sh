cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $G/prof_blk
rocprofv3 --kernel-trace --stats -d $G/prof_blk -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --blocking --no-legs > $G/blk.log 2>&1
