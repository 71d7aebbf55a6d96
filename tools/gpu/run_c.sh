python tools/gpu/determinism.py 2>&1 | grep -v amdgpu | tail -12
