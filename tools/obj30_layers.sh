cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
for o in 1 30; do
rm -rf $G/lt_o$o
(cd $R && rocprofv3 --kernel-trace -d $G/lt_o$o -o t -- python bench.py --steps 2 --warmup 1 --blocking --no-legs --objects $o > /dev/null 2>&1)
python $R/tools/layer_times.py $(find $G/lt_o$o -name "t_results.db" | head -1) > $G/layers_obj$o.txt 2>&1
python $R/tools/rocprof_summary.py $(find $G/lt_o$o -name "t_results.db" | head -1) > $G/stats_obj$o.txt 2>&1
rm -rf $G/lt_o$o
done
paste <(awk '{print $1, $(NF-7)}' $G/layers_obj1.txt) <(awk '{print $(NF-7)}' $G/layers_obj30.txt) | head -50
