timeout 1200 python -m pytest tests/test_pnp_gpu.py tests/test_est_pose_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -5
python tools/single_det.py 100
