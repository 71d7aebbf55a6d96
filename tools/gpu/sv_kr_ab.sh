for V in sv_kr4 sv_kr5 sv_kr6; do echo "== tests $V"; F4L_LIB_PATH=$PWD/fusion4landslide_amd/lib/variants/lib_$V.so timeout 300 python -m pytest tests/test_gpu_supervoxel_parallel.py -x -q -m gpu 2>&1 | tail -1; done
TAIL=1 timeout 400 bash tools/gpu/lib_ab.sh "python3 tools/gpu/svp_only.py 10000000 3" sv_kr4 sv_kr5 sv_kr6
TAIL=1 timeout 400 bash tools/gpu/lib_ab.sh "python3 tools/gpu/sv_only.py 1000000 0.52 4" sv_kr4 sv_kr5 sv_kr6
TAIL=1 timeout 400 bash tools/gpu/lib_ab.sh "python3 tools/gpu/sv_only.py 1000000 1.386 4" sv_kr4 sv_kr5 sv_kr6
