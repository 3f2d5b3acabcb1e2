#!/bin/bash
# Regenerate the evidence under profiles/ (run from the repo root in the build container).
#   1. on the GPU box: full GPU test suite, rocprofv3 kernel trace of `bench.py --blocking` (one batch on the GPU at a time,
#      so kernel durations are the kernels' own; the default stream mode overlaps two batches), two PMC passes
#      (FETCH_SIZE / WRITE_SIZE separately, MI355X_MICROARCH.md), and a plain bench run
#   2. here: summarise the rocpd databases into profiles/
set -e
R=${1:-r01}
/usr/local/graft/bin/gpurun --timeout 2400 -- '
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/gpu_tests.log
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $G/prof_final $G/pmc_fetch2 $G/pmc_write2
rocprofv3 --kernel-trace --stats -d $G/prof_final -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --blocking > $G/bench_final_prof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $G/pmc_fetch2 -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --blocking > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $G/pmc_write2 -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --blocking > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py gpurun_out/pmc_fetch2/x_results.db gpurun_out/pmc_write2/x_results.db profiles/'"$R"'_traffic.json > /dev/null   # bench.py reads it
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python bench.py --steps 5 --warmup 2 --objects 30 --cpu-sample 0 > gpurun_out/bench_objects30.json 2>> gpurun_out/bench_final.err
python bench.py --steps 5 --warmup 2 --precision f32 --cpu-sample 0 > gpurun_out/bench_f32.json 2>> gpurun_out/bench_final.err
cat gpurun_out/gpu_tests.log; tail -c 400 gpurun_out/bench_final.json
' 2>&1 | tail -8
python tools/rocprof_summary.py gpurun_out/prof_final/bench_results.db > profiles/${R}_bench_kernel_stats.txt
python tools/layer_times.py gpurun_out/prof_final/bench_results.db > profiles/${R}_layer_times.txt
python tools/pmc_traffic.py gpurun_out/pmc_fetch2/x_results.db gpurun_out/pmc_write2/x_results.db profiles/${R}_traffic.json | head -4
cp gpurun_out/bench_final.json profiles/${R}_bench_line.json
cp gpurun_out/bench_objects30.json profiles/${R}_bench_line_objects30.json
cp gpurun_out/bench_f32.json profiles/${R}_bench_line_f32mode.json
cp gpurun_out/gpu_tests.log profiles/${R}_gpu_tests.log
head -14 profiles/${R}_bench_kernel_stats.txt | cut -c1-175
