cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
for v in head v2 v1 nomm; do
  e="P2P_AB=1"; [ $v != head ] && e="P2P_LIB=$R/tools/ab/aa_$v/libp2p_mi355.so"
  rm -rf $G/prof_aav
  (cd $R && env $e rocprofv3 --kernel-trace --stats -d $G/prof_aav -o bench -- python bench.py --steps 3 --warmup 1 --blocking --no-legs --bbox-side 40,300 --anti-aliasing > /dev/null 2>&1)
  echo "== $v"; python $R/tools/rocprof_summary.py $(find $G/prof_aav -name "bench_results.db" | head -1) | grep "aa_filter" | cut -c1-80,108-175
done
rm -rf $G/prof_aav
