#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer build of libp2p_mi355.so (sanitizer builds of
# the host code -- grow-only workspaces, slot reuse across tickets, pinned landing buffers, marshalling).  Device code is
# compiled as usual (-Xarch_host limits the instrumentation to the host pass).  Run from the repo root:
#     tools/sanitize.sh build           # here (hipcc cross-compiles)            -> tools/ab/libp2p_asan.so
#     tools/sanitize.sh run [pytest args]   # on the GPU box: runs GPU tests against the instrumented library
set -e
cd "$(dirname "$0")/.."
OUT=tools/ab/libp2p_asan.so
MODE=${SAN_MODE:-asan}          # asan: AddressSanitizer + UBSan; ubsan: UBSan + _GLIBCXX_ASSERTIONS only (no runtime interceptors)
if [ "$MODE" = "ubsan" ]; then OUT=tools/ab/libp2p_ubsan.so; fi
if [ "$1" = "build" ]; then
    mkdir -p tools/ab /tmp/p2p_asan
    python -c "from pix2pose_amd import build; build.build()"     # regenerates csrc/_build_id.cpp
    OBJS=""
    SRCS=$(python -c "from pix2pose_amd import build; print(' '.join(s[:-4] for s in build.SOURCES))")      # every source of the library
    if [ "$MODE" = "ubsan" ]; then SAN="-Xarch_host -fsanitize=undefined -Xarch_host -D_GLIBCXX_ASSERTIONS"; else SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined"; fi
    if [ "$MODE" = "ubsan" ]; then LSAN="-fsanitize=undefined"; else LSAN="-fsanitize=address -fsanitize=undefined"; fi
    g++ -O1 -fPIC -c pix2pose_amd/csrc/_build_id.cpp -o /tmp/p2p_asan/_build_id.o
    # twice: the shipped configuration, and the development twin (-DP2P_DEV_SWITCHES) the route-equivalence tests load through
    # build.dev_switches() (P2P_DEV_LIB names the instrumented twin for them)
    for variant in "" dev; do
        OBJS=""; n=0
        for s in $SRCS; do
            /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC $SAN ${variant:+-DP2P_DEV_SWITCHES} \
                -Xarch_host -fno-omit-frame-pointer -Xarch_host -fno-sanitize-recover=undefined -c pix2pose_amd/csrc/$s.hip -o /tmp/p2p_asan/$s$variant.o &
            OBJS="$OBJS /tmp/p2p_asan/$s$variant.o"
            n=$((n + 1)); if [ $((n % 4)) -eq 0 ]; then wait; fi
        done
        wait
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $LSAN -shared-libsan -o ${OUT%.so}${variant:+_dev}.so $OBJS /tmp/p2p_asan/_build_id.o -ldl
        echo "built ${OUT%.so}${variant:+_dev}.so"
    done
    exit 0
fi
shift || true
if [ "$MODE" = "ubsan" ]; then
    RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
    export MALLOC_CHECK_=3 MALLOC_PERTURB_=165         # glibc heap consistency checks + poisoned fresh / freed memory
else
    RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
fi
# python itself is not instrumented: preload the runtime; leaks are not checked (the interpreter and HIP keep process-lifetime
# allocations); the library is named through P2P_LIB (the binding then skips its build-id check)
LD_PRELOAD=$RT ASAN_OPTIONS=${ASAN_OPTIONS:-detect_leaks=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0} UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    P2P_LIB=$PWD/$OUT P2P_DEV_LIB=$PWD/${OUT%.so}_dev.so python -m pytest tests -m gpu -q -x -p no:cacheprovider \
    --deselect tests/test_bench_multirank_gpu.py "$@"
