#!/bin/bash
# configs[1] leg (64-input generator pass) under route thresholds of the development twin: tools/batch64_ab.sh
cd $GRAFT_REPO_ROOT
L=$PWD/pix2pose_amd/libp2p_mi355_dev.so
run() {   # label, env...
    local lab=$1; shift
    env P2P_LIB=$L "$@" python bench.py --steps 2 --warmup 1 --f32-steps 0 --host-frames 0 --latency 0 --cpu-sample 0 --general 0 --batch64 30 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read())['batch64']; print('%-28s' % '$lab', ' '.join('%s %.0f' % (k, d[k]['inputs_per_s']) for k in ('n64','n256','n768')))"
}
for r in 1 2; do
run default P2P_AB=1
run stream128 P2P_STREAM_WGS=128 P2P_FUSED_MIN_WGS=65
run stream256 P2P_STREAM_WGS=256 P2P_FUSED_MIN_WGS=65
run stream512 P2P_STREAM_WGS=512 P2P_FUSED_MIN_WGS=65
done
