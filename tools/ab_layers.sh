#!/bin/bash
# Per-layer times of the stage-1 pass for a list of A/B library builds (run on the GPU box):  tools/ab_layers.sh name=path ...
# ("base" = the tree's own library).  Output: gpurun_out/ab_<name>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    rm -rf $R/gpurun_out/ab_prof_$name
    if [ "$lib" == "base" ]; then unset P2P_LIB; else export P2P_LIB=$R/$lib; fi
    rocprofv3 --kernel-trace -d $R/gpurun_out/ab_prof_$name -o t -- python $R/bench.py --steps 2 --warmup 1 --blocking --no-legs > /dev/null 2>&1
    python $R/tools/layer_times.py $R/gpurun_out/ab_prof_$name/t_results.db > $R/gpurun_out/ab_$name.txt
    rm -rf $R/gpurun_out/ab_prof_$name
    echo "== $name: $(tail -1 $R/gpurun_out/ab_$name.txt)"
done
