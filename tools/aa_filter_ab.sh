#!/bin/bash
# GPU box: the anti-aliasing filter A/B -- tools/ab/aaold/libp2p_mi355.so (this tree with the previous resize_aa.hip, built by hand: see the round-5
# notes in tools/experiments/README.md) against this tree; general-crop legs, one lease, interleaved.  Output: gpurun_out/r05_aa_filter_ab.txt
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
REP=$G/r05_aa_filter_ab.txt
echo "# anti-aliasing filter: old = one output per thread, new = 8 (rows) / 4 (columns) outputs per thread from one register window; bench.py --steps 10 --general 10, other legs off; same lease, interleaved" > $REP
for r in 1 2; do
  for v in old new; do
    e="P2P_AB=1"; [ $v == old ] && e="P2P_LIB=$R/tools/ab/aaold/libp2p_mi355.so"
    line=$(cd $R && env $e python bench.py --steps 10 --warmup 2 --f32-steps 0 --host-frames 0 --latency 0 --cpu-sample 0 --general 10 --batch64 0 2>/dev/null | tail -1)
    python - "$v" "$line" >> $REP <<'EOF'
import json, sys
d = json.loads(sys.argv[2]); g = d["general_crops"]
print("%-4s headline %7.1f   general %7.1f (%.3f of headline)   with anti-aliasing %7.1f (%.3f)" % (sys.argv[1], d["value"], g["value"], g["value"] / d["value"], g["value_anti_aliasing"], g["value_anti_aliasing"] / d["value"]))
EOF
  done
done
rm -rf $G/prof_gaa
(cd $R && rocprofv3 --kernel-trace --stats -d $G/prof_gaa -o bench -- python bench.py --steps 3 --warmup 1 --blocking --no-legs --bbox-side 40,300 --anti-aliasing > /dev/null 2>&1)
python $R/tools/rocprof_summary.py $(find $G/prof_gaa -name "bench_results.db" | head -1) > $G/r05_general_crops_aa_kernel_stats.txt
rm -rf $G/prof_gaa
cat $REP; grep "aa_" $G/r05_general_crops_aa_kernel_stats.txt | cut -c1-80,108-175
