python -m pytest tests -m gpu -q -x -k "dense or shortcuts or full_size" 2>&1 | tail -15
