#!/bin/bash
# GPU box: the general-crop-size leg (bbox sides 40 - 300 px, with and without the 0.17 / 0.18 resize generation) of prev (tools/ab/prev) and head on ONE lease,
# interleaved, + the kernel stats of one blocking run of head with the filter.  Output: gpurun_out/r05_general_ab.txt
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
REP=$G/r05_general_ab.txt
echo "# general_crops leg (256 detections / step, bbox sides U(40, 300) px), bench.py --steps 10 --general 10, other legs off; same lease, interleaved" > $REP
for r in 1 2; do
  for v in prev head; do
    tree=$R; extra="--batch64 0"; [ $v == prev ] && tree=$R/tools/ab/prev && extra=""
    line=$(cd $tree && python bench.py --steps 10 --warmup 2 --f32-steps 0 --host-frames 0 --latency 0 --cpu-sample 0 --general 10 $extra 2>/dev/null | tail -1)
    python - "$v" "$line" >> $REP <<'EOF'
import json, sys
d = json.loads(sys.argv[2]); g = d["general_crops"]
print("%-5s headline %7.1f   general %7.1f (%.3f of headline)   with anti-aliasing %7.1f (%.3f)" % (sys.argv[1], d["value"], g["value"], g["value"] / d["value"], g["value_anti_aliasing"], g["value_anti_aliasing"] / d["value"]))
EOF
  done
done
rm -rf $G/prof_gaa
(cd $R && rocprofv3 --kernel-trace --stats -d $G/prof_gaa -o bench -- python bench.py --steps 3 --warmup 1 --blocking --no-legs --bbox-side 40,300 --anti-aliasing > /dev/null 2>&1)
python $R/tools/rocprof_summary.py $(find $G/prof_gaa -name "bench_results.db" | head -1) > $G/r05_general_crops_aa_kernel_stats.txt
rm -rf $G/prof_gaa
cat $REP; head -24 $G/r05_general_crops_aa_kernel_stats.txt | cut -c1-80,108-175
