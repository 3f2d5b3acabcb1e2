timeout 1700 python -m pytest tests/test_est_pose_gpu.py tests/test_golden_gpu.py tests/test_reference_vectors_gpu.py tests/test_misc_gpu.py tests/test_eval_bop.py -x -q -m gpu 2>&1 | tail -3
python tools/single_det.py 200 | tail -1
python tools/single_det.py 200 | tail -1
