#!/bin/bash
# A/B build of the library with one source recompiled under an extra macro: tools/ab_build.sh <source.hip> <-DMACRO> -> tools/ab/libp2p_ab.so
# (load it with P2P_LIB=tools/ab/libp2p_ab.so; the main build's objects must exist: python pix2pose_amd/build.py)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/pix2pose_amd/csrc
mkdir -p $ROOT/tools/ab
SRC=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $C/$SRC -o $ROOT/tools/ab/ab_${SRC%.hip}.o 2>&1 | grep -v "warning\|^ \|^$" || true
OBJS=""
for f in $C/*.o; do
    b=$(basename $f .o)
    if [ "$b" == "${SRC%.hip}" ]; then OBJS="$OBJS $ROOT/tools/ab/ab_${SRC%.hip}.o"; else OBJS="$OBJS $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/ab/libp2p_ab.so $OBJS
echo $ROOT/tools/ab/libp2p_ab.so
