timeout 1500 python -m pytest tests/test_pnp_gpu.py tests/test_est_pose_gpu.py tests/test_golden_gpu.py tests/test_reference_vectors_gpu.py -x -q 2>&1 | tail -3
python tools/single_det.py 200 | tail -1
python tools/single_det.py 200 | tail -1
