for v in "0 4" "4 4" "0 2" "0 1" "0 4"; do set -- $v; echo "debug=$1 waves=$2"; F4L_ICP_DEBUG=$1 F4L_ICP_WAVES=$2 python -m pytest tests -m gpu -q -x -k "point2plane" 2>&1 | tail -3; done
