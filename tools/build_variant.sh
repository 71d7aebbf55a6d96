#!/bin/bash
# Usage: tools/build_variant.sh <name> [make variables...]   e.g.  tools/build_variant.sh icp_prof PROF=1
# Builds the library from the CURRENT sources in a scratch copy of csrc/ and leaves it as fusion4landslide_amd/lib/variants/lib_<name>.so
# (git-ignored; travels to the GPU box; load it with F4L_LIB_PATH) -- A/B and profiling builds beside the product build.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
B="$ROOT/build/variant_$name"   # (/build/ is git-ignored and gpurun-ignored)
mkdir -p "$B/fusion4landslide_amd/csrc" "$B/include"
cp "$ROOT"/fusion4landslide_amd/csrc/*.hip "$ROOT"/fusion4landslide_amd/csrc/*.h "$ROOT"/fusion4landslide_amd/csrc/*.cpp "$ROOT"/fusion4landslide_amd/csrc/Makefile "$B/fusion4landslide_amd/csrc/"
cp "$ROOT"/include/*.h "$B/include/"
make -C "$B/fusion4landslide_amd/csrc" -j6 "$@" > "$B/build.log" 2>&1 || { tail -30 "$B/build.log"; exit 1; }
mkdir -p "$ROOT/fusion4landslide_amd/lib/variants"
cp "$B/fusion4landslide_amd/lib/libf4l_hip.so" "$ROOT/fusion4landslide_amd/lib/variants/lib_$name.so"
cp "$B/build.log" "$ROOT/build/variant_$name.log"; rm -rf "$B"   # (the scratch copy does not travel to the GPU box with the next snapshot)
echo "built fusion4landslide_amd/lib/variants/lib_$name.so"
