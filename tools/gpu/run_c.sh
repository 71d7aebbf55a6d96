python -m pytest tests -m gpu -q -x -k "robust_rigid or kabsch2" 2>&1 | tail -12
