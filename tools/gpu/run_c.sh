python -m pytest tests -m gpu -q -x -k "voxel or tiling" 2>&1 | tail -8
