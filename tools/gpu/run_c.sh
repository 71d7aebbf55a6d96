python -m pytest tests -m gpu -q -x -k "full_size" 2>&1 | grep -E "assert|passed|failed|dev32" | head
