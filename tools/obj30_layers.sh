#!/bin/bash
# GPU box: kernel stats + per-layer times of one blocking step with 1 and with 30 object models (BASELINE.json configs[3])
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
for o in 1 30; do
rm -rf $G/lt_o$o
(cd $R && rocprofv3 --kernel-trace -d $G/lt_o$o -o t -- python bench.py --steps 2 --warmup 1 --blocking --no-legs --objects $o > /dev/null 2>&1)
python $R/tools/layer_times.py $(find $G/lt_o$o -name "t_results.db" | head -1) > $G/layers_obj$o.txt 2>&1
python $R/tools/rocprof_summary.py $(find $G/lt_o$o -name "t_results.db" | head -1) > $G/stats_obj$o.txt 2>&1
rm -rf $G/lt_o$o
done
head -24 $G/stats_obj30.txt | cut -c1-80,108-175
