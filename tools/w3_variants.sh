#!/bin/bash
# GPU box: wino3_gemm / wino3_input time per pass for A/B builds: [N=256] tools/w3_variants.sh name ...  (libs tools/ab/libp2p_<name>.so; "base" = the tree's library)
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for name in "$@"; do
    if [ "$name" == "base" ]; then unset P2P_LIB; else export P2P_LIB=$R/tools/ab/libp2p_$name.so; fi
    echo "$name: $(python $R/tools/time_pass.py ${N:-256} 20 resnet50 auto 2>&1 | tail -1)"
done
done
