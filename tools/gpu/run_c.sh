python -m pytest tests -m gpu -q 2>&1 | tail -5
python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --extras 0 | tail -1 | cut -c1-330
