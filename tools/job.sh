#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_wino_gpu.py tests/test_ae_gpu.py tests/test_stream_gpu.py tests/test_fullsize_gpu.py tests/test_reference_vectors_gpu.py -x -q -m gpu 2>&1 | tail -2
python tools/time_small.py resnet50 50 2>&1 | grep ms/pass | awk '{printf "%s %s  ", $2, $3} END{print ""}'
