#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu/lib_ab.sh "<command>" <variant> [<variant> ...]  -- the command under the product library
# ("product") and under fusion4landslide_amd/lib/variants/lib_<variant>.so (tools/build_variant.sh), two rounds so that drift shows.
CMD="$1"; shift
for rep in 1 2; do
  for V in product "$@"; do
    if [ "$V" = product ]; then L=""; else L="$PWD/fusion4landslide_amd/lib/variants/lib_$V.so"; fi
    echo "== $V (round $rep)"
    F4L_LIB_PATH=$L $CMD 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-2}
  done
done
