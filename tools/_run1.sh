timeout 1700 python -m pytest tests/test_bench_multirank_gpu.py -x -q 2>&1 | tail -8
python bench.py --steps 10 --warmup 3 > gpurun_out/bench_r3a.json 2> gpurun_out/bench_r3a.err; tail -c 600 gpurun_out/bench_r3a.err
