for b in 0 512 256 128; do echo BIG $b; P2P_STREAM_BIG=$b python tools/time_small.py resnet50 50 1,3,8 2>&1 | grep resnet; done
timeout 600 env P2P_STREAM_BIG=256 python -m pytest tests/test_stream_gpu.py -x -q 2>&1 | tail -2
