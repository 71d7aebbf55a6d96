python -m pytest tests -m gpu -q -x -k "full_size" 2>&1 | tail -30
