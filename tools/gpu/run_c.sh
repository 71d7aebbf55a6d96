python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python tools/gpu/time_knn.py 2>&1 | grep -v amdgpu
