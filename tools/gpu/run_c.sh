python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -12
