#!/bin/bash
# GPU box: PnP hypothesis rounds, coarse ([0,16) [16,64) [64,100)) against fine ([16,32) [32,48) [48,64) in place of the second) for batches --
# the development twin with P2P_PNP_FINE_MIN; headline + general-crop legs, one lease, interleaved.  Output: gpurun_out/r05_pnp_rounds_ab.txt
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
REP=$G/r05_pnp_rounds_ab.txt
echo "# PnP hypothesis rounds: coarse = P2P_PNP_FINE_MIN=1000000 (three rounds for every launch), fine = default (five rounds from 17 problems); development twin, bench.py --steps 10 --general 10, other legs off" > $REP
for r in 1 2; do
  for v in coarse fine; do
    e="P2P_AB=1"; [ $v == coarse ] && e="P2P_PNP_FINE_MIN=1000000"
    line=$(cd $R && env P2P_LIB=$R/pix2pose_amd/libp2p_mi355_dev.so $e python bench.py --steps 10 --warmup 2 --f32-steps 0 --host-frames 0 --latency 0 --cpu-sample 0 --general 10 --batch64 0 2>/dev/null | tail -1)
    python - "$v" "$line" >> $REP <<'EOF'
import json, sys
d = json.loads(sys.argv[2]); g = d["general_crops"]
print("%-6s headline %7.1f   general %7.1f (%.3f of headline)   with anti-aliasing %7.1f (%.3f)" % (sys.argv[1], d["value"], g["value"], g["value"] / d["value"], g["value_anti_aliasing"], g["value_anti_aliasing"] / d["value"]))
EOF
  done
done
cat $REP
