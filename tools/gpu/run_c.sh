python -m pytest tests -m gpu -q -x -k "workgroup_shape or medium_patches" 2>&1 | tail -15
