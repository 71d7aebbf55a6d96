python -m pytest tests -m gpu -q -x -k "kabsch or median" 2>&1 | tail -15
