python -m pytest tests -m gpu -q -x -k "nn_query or voxel or knn or median" --durations=6 2>&1 | tail -12
python tools/gpu/time_knn.py 2>&1 | tail -3
