#!/bin/bash
# wino3.hip (transposed convolutions in Winograd F(4,3) form) against the direct phase kernels, same lease, interleaved:
#   tools/wino3_ab.sh   (on the GPU box, from the repo root) -> gpurun_out/wino3_ab.txt
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out; O=$G/wino3_ab.txt
DEV=$R/pix2pose_amd/libp2p_mi355_dev.so
: > $O
for rep in 1 2; do
for n in 256 64 768; do
    echo "head      $(python $R/tools/time_pass.py $n 20 resnet50 auto 2>&1 | tail -1)" >> $O
    echo "no wino3  $(P2P_LIB=$DEV P2P_NO_WINO3=1 python $R/tools/time_pass.py $n 20 resnet50 auto 2>&1 | tail -1)" >> $O
done
done
for n in 2 3 4 8 16 32; do
    echo "always    $(python $R/tools/time_pass.py $n 50 resnet50 always 2>&1 | tail -1)" >> $O
    echo "no wino3  $(P2P_LIB=$DEV P2P_NO_WINO3=1 python $R/tools/time_pass.py $n 50 resnet50 always 2>&1 | tail -1)" >> $O
done
cat $O
