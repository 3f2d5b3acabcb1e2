#!/bin/bash
# Regenerate the evidence under profiles/ (run from the repo root in the build container).
#   1. on the GPU box: full GPU test suite, rocprofv3 kernel trace of `bench.py --blocking` (one batch on the GPU at a time,
#      so kernel durations are the kernels' own; the default stream mode overlaps two batches), PMC passes -- HBM traffic
#      (FETCH_SIZE / WRITE_SIZE separately, MI355X_MICROARCH.md) and SQ counters in three passes (never together with a
#      --sys-trace / HIP trace domain) -- and the bench lines (default, 30 objects, strict fp32, 2 ranks on one device)
#   2. here: summarise the rocpd databases into profiles/
set -e
R=${1:-r06}
PROF="--steps 3 --warmup 1 --blocking --no-legs"
PMC="--steps 1 --warmup 1 --blocking --no-legs"
/usr/local/graft/bin/gpurun --timeout 2400 -- '
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/gpu_tests.log
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
B=$GRAFT_REPO_ROOT/bench.py
rm -rf $G/prof_final $G/prof_general $G/prof_general_aa $G/pmc_fetch2 $G/pmc_write2 $G/pmc_sq1 $G/pmc_sq2 $G/pmc_sq3 $G/prof_call $G/prof_small1 $G/prof_small3
rocprofv3 --kernel-trace -d $G/prof_call -o t -- python $GRAFT_REPO_ROOT/tools/single_det.py 10 > $G/call.log 2>&1
rocprofv3 --kernel-trace -d $G/prof_small1 -o t -- python $GRAFT_REPO_ROOT/tools/time_small.py resnet50 3 1 > /dev/null 2>&1
rocprofv3 --kernel-trace -d $G/prof_small3 -o t -- python $GRAFT_REPO_ROOT/tools/time_small.py resnet50 3 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $G/prof_final -o bench -- python $B '"$PROF"' > $G/bench_final_prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $G/prof_general -o bench -- python $B '"$PROF"' --bbox-side 40,300 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $G/prof_general_aa -o bench -- python $B '"$PROF"' --bbox-side 40,300 --anti-aliasing > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $G/pmc_fetch2 -o x -- python $B '"$PMC"' > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $G/pmc_write2 -o x -- python $B '"$PMC"' > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $G/pmc_sq1 -o x -- python $B '"$PMC"' > $G/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d $G/pmc_sq2 -o x -- python $B '"$PMC"' > $G/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS -d $G/pmc_sq3 -o x -- python $B '"$PMC"' > $G/pmc_sq3.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py gpurun_out/pmc_fetch2/x_results.db gpurun_out/pmc_write2/x_results.db profiles/'"$R"'_traffic.json > /dev/null   # bench.py reads it
python tools/single_det.py 200 > gpurun_out/single_det.log 2>&1
python tools/time_small.py resnet50 50 > gpurun_out/small_passes.log 2>&1
P2P_LIB=$GRAFT_REPO_ROOT/pix2pose_amd/libp2p_mi355_dev.so P2P_STREAM_WGS=0 python tools/time_small.py resnet50 50 1,3,8 >> gpurun_out/small_passes.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python bench.py --steps 5 --warmup 2 --objects 30 --no-legs > gpurun_out/bench_objects30.json 2>> gpurun_out/bench_final.err
python bench.py --steps 5 --warmup 2 --precision f32 --no-legs > gpurun_out/bench_f32.json 2>> gpurun_out/bench_final.err
python bench.py --steps 5 --warmup 2 --masks --no-legs > gpurun_out/bench_masks.json 2>> gpurun_out/bench_final.err
python bench.py --gpus 2 --backend gloo --same-device --steps 3 --warmup 1 --no-legs > gpurun_out/bench_2ranks_same_device.json 2>> gpurun_out/bench_final.err
python bench.py --steps 20 --warmup 3 --no-legs > gpurun_out/bench_k20.json 2>> gpurun_out/bench_final.err
python bench.py --backbone paper --steps 20 --warmup 3 --no-legs > gpurun_out/bench_paper.json 2>> gpurun_out/bench_final.err
timeout 300 tools/probe_kernel.sh resblock 256 > /dev/null 2>&1
[ -d tools/ab/prev ] && timeout 600 tools/ab_round.sh run 3 > gpurun_out/ab_round.log 2>&1      # the A/B worktree travels only when .gpurunignore lets it
[ -d tools/ab/prev ] && timeout 300 tools/general_ab.sh > /dev/null 2>&1
timeout 200 tools/batch_layers.sh 64 > /dev/null 2>&1
(python tools/soak_est_pose.py 60 2 7000; python tools/soak_est_pose.py 40 1 7100; python tools/soak_est_pose.py 25 0 7200; python tools/soak_est_pose.py 8 0 7300 40; python tools/soak_est_pose.py 60 0 7400 1) 2>/dev/null | grep -E "^soak|MISMATCH" > gpurun_out/soak.txt
cat gpurun_out/gpu_tests.log; tail -c 300 gpurun_out/bench_final.json
' 2>&1 | tail -8
python tools/rocprof_summary.py gpurun_out/prof_final/bench_results.db > profiles/${R}_bench_kernel_stats.txt
python tools/layer_times.py gpurun_out/prof_final/bench_results.db > profiles/${R}_layer_times.txt
python tools/rocprof_summary.py gpurun_out/prof_general/bench_results.db > profiles/${R}_general_crops_kernel_stats.txt
python tools/rocprof_summary.py gpurun_out/prof_general_aa/bench_results.db > profiles/${R}_general_crops_aa_kernel_stats.txt
python tools/pmc_traffic.py gpurun_out/pmc_fetch2/x_results.db gpurun_out/pmc_write2/x_results.db profiles/${R}_traffic.json | head -4
python tools/pmc_sq.py gpurun_out/pmc_sq1/x_results.db gpurun_out/pmc_sq2/x_results.db gpurun_out/pmc_sq3/x_results.db > profiles/${R}_sq_counters.txt
python tools/trace_call.py gpurun_out/prof_call/t_results.db > profiles/${R}_single_det_trace.txt
python tools/trace_small.py gpurun_out/prof_small1/t_results.db > profiles/${R}_small_pass_n1.txt
python tools/trace_small.py gpurun_out/prof_small3/t_results.db > profiles/${R}_small_pass_n3.txt
grep -h "single est_pose\|C call" gpurun_out/single_det.log > profiles/${R}_single_det.txt
grep -h resnet50 gpurun_out/small_passes.log > profiles/${R}_small_passes.txt
for f in bench_final:bench_line bench_objects30:bench_line_objects30 bench_f32:bench_line_f32mode bench_masks:bench_line_masks bench_2ranks_same_device:bench_line_2ranks_same_device bench_k20:bench_line_k20 bench_paper:bench_line_paper_backbone; do
    grep '^{' gpurun_out/${f%%:*}.json | tail -1 > profiles/${R}_${f##*:}.json
done
cp gpurun_out/gpu_tests.log profiles/${R}_gpu_tests.log
# the real libraries of the image's second interpreter (scikit-image 0.18.3, h5py 3.3.0): HDF5 reader on real files, fixtures re-derived
(echo "# /opt/conda/bin/python3.9 -m pytest tests/test_convert_keras.py"; PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore -m pytest tests/test_convert_keras.py -q -p no:cacheprovider 2>&1 | tail -3
 echo "# python -m pytest tests/test_real_libraries_cpu.py tests/test_external_vectors.py tests/test_reference_vectors_cpu.py"; python -m pytest tests/test_real_libraries_cpu.py tests/test_external_vectors.py tests/test_reference_vectors_cpu.py -q -m "not gpu" 2>&1 | tail -3) > profiles/${R}_real_libraries.log
for f in layers_b64:layer_times_b64 probe_resblock:resblock_counters ${R}_vs_prev:vs_prev ${R}_general_ab:general_ab ${R}_general_crops_aa_kernel_stats:general_crops_aa_kernel_stats_ab soak:soak; do
    [ -f gpurun_out/${f%%:*}.txt ] && cp gpurun_out/${f%%:*}.txt profiles/${R}_${f##*:}.txt
done
[ -f gpurun_out/pnp_exact_match.json ] && cp gpurun_out/pnp_exact_match.json profiles/${R}_pnp_exact_match.json
[ -f gpurun_out/precision_report.json ] && cp gpurun_out/precision_report.json profiles/${R}_precision_report.json
head -14 profiles/${R}_bench_kernel_stats.txt | cut -c1-175
