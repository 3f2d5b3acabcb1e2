set -x
timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_ae_gpu.py -x -q 2>&1 | tail -15
timeout 300 python tools/time_small.py resnet50 50 2>&1 | tail -12
P2P_STREAM_WGS=0 timeout 300 python tools/time_small.py resnet50 50 2>&1 | tail -12
