#!/bin/bash
# GPU box: per-layer times of one blocking 256-detection step (rocprofv3 kernel trace + tools/layer_times.py) -> gpurun_out/layers_<tag>.txt
TAG=${1:-head}; shift
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $G/lt_prof
(cd $R && env "$@" rocprofv3 --kernel-trace -d $G/lt_prof -o t -- python bench.py --steps 2 --warmup 1 --blocking --no-legs > /dev/null 2>&1)
python $R/tools/layer_times.py $(find $G/lt_prof -name "t_results.db" | head -1) > $G/layers_$TAG.txt 2>&1
rm -rf $G/lt_prof
head -${LINES_SHOWN:-12} $G/layers_$TAG.txt; tail -1 $G/layers_$TAG.txt
