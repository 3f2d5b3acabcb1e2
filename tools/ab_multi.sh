#!/bin/bash
# Several A/B builds of the library at once, each with ONE source recompiled under extra macros:
#   tools/ab_multi.sh <source.hip> name1="-DMACRO_A" name2="-DMACRO_B -DMACRO_C" ...   ->  tools/ab/libp2p_<name>.so
# (the main build's objects must exist: python pix2pose_amd/build.py).  Run them on the GPU box with tools/ab_layers.sh name=tools/ab/libp2p_<name>.so ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/pix2pose_amd/csrc
mkdir -p $ROOT/tools/ab
SRC=$1; shift
for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $C/$SRC -o $ROOT/tools/ab/${name}_${SRC%.hip}.o 2>&1 | grep -v "warning\|^ \|^$" || true
    OBJS=""
    for f in $C/*.o; do
        b=$(basename $f .o)
        if [ "$b" == "${SRC%.hip}" ]; then OBJS="$OBJS $ROOT/tools/ab/${name}_${SRC%.hip}.o"; else OBJS="$OBJS $f"; fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/ab/libp2p_$name.so $OBJS
    echo tools/ab/libp2p_$name.so
done
