python -m pytest tests -m gpu -q 2>&1 | tail -5
python bench.py --steps 10 --warmup 3 --cpu-seconds 0 2>&1 | tail -1
