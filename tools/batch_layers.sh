#!/bin/bash
# Per-layer times of one blocking generator pass at a given batch (rocprofv3 kernel trace): tools/batch_layers.sh 64 [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out; B=${1:-64}; shift
rm -rf $G/lt_b$B
(cd $R && rocprofv3 --kernel-trace -d $G/lt_b$B -o t -- python bench.py --steps 2 --warmup 1 --blocking --no-legs --batch $B "$@" > /dev/null 2>&1)
python $R/tools/layer_times.py $(find $G/lt_b$B -name "t_results.db" | head -1) $B > $G/layers_b$B.txt 2>&1
rm -rf $G/lt_b$B
cat $G/layers_b$B.txt
