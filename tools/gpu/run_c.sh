python -m pytest tests -m gpu -q -x -k "size_classes" 2>&1 | tail -8
