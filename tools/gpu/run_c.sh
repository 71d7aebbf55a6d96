python tools/gpu/scale_p.py 64 256 512 1024 1536 2025 2>&1 | tail -7
