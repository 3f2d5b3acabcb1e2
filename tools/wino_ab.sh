#!/bin/bash
# GPU box: per-layer times of the Winograd layers for a list of A/B builds: tools/wino_ab.sh [batch] name=lib ...   (base = the tree's library)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; B=$1; shift
for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    rm -rf $R/gpurun_out/ab_prof_$name
    if [ "$lib" == "base" ]; then unset P2P_LIB; else export P2P_LIB=$R/$lib; fi
    rocprofv3 --kernel-trace -d $R/gpurun_out/ab_prof_$name -o t -- python $R/bench.py --steps 2 --warmup 1 --blocking --no-legs --batch $B > /dev/null 2>&1
    python $R/tools/layer_times.py $R/gpurun_out/ab_prof_$name/t_results.db $B > $R/gpurun_out/ab_$name.txt
    rm -rf $R/gpurun_out/ab_prof_$name
    echo "== $name: $(grep deconv $R/gpurun_out/ab_$name.txt | awk '{printf "%s %s %s us | ", $1, $2, $(NF-7)}') $(tail -1 $R/gpurun_out/ab_$name.txt)"
done
