cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
cd $R && python -m pytest tests -m gpu -q -x -k "resize or aa or anti or est_pose or skimage or reference_vectors" 2>&1 | tail -2
cd /tmp
for side in 40,300 250,420; do
for v in aaold head; do
  e="P2P_AB=1"; [ $v != head ] && e="P2P_LIB=$R/tools/ab/$v/libp2p_mi355.so"
  rm -rf $G/prof_aav
  (cd $R && env $e rocprofv3 --kernel-trace --stats -d $G/prof_aav -o bench -- python bench.py --steps 3 --warmup 1 --blocking --no-legs --bbox-side $side --anti-aliasing > $G/aa_big_$v.log 2>&1)
  echo "== $v $side $(tail -1 $G/aa_big_$v.log | python -c 'import json,sys; print(json.loads(sys.stdin.read())["value"])' 2>/dev/null)"; python $R/tools/rocprof_summary.py $(find $G/prof_aav -name "bench_results.db" | head -1) | grep "aa_filter" | cut -c1-80,108-175
done
done
rm -rf $G/prof_aav
