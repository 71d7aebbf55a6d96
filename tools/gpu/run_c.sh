python -m pytest tests -m gpu -q -x -k "supervoxel" 2>&1 | tail -5
F4L_SV_TIMING=1 python tools/gpu/time_knn.py 2>&1 | tail -3
F4L_SV_HOST_ONLY=1 F4L_SV_TIMING=1 python tools/gpu/time_knn.py 2>&1 | tail -2
