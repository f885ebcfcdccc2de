#!/bin/bash
# A build of the library with extra compiler flags, for A/B runs through SG_HIP_LIB (scripts/gpu_session.sh ab:):
#   scripts/build_variant.sh <name> <flags...>   ->  string_grouper_amd/libsg_hip_<name>.so   (git-ignored; travels with gpurun)
# Built in a scratch copy of the sources, so the objects of the real build stay as they are.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
work=${TMPDIR:-/tmp}/sg_variant_$name
rm -rf "$work" && mkdir -p "$work/string_grouper_amd" "$work/include"
cp -r "$root/string_grouper_amd/csrc" "$work/string_grouper_amd/"
cp "$root"/include/*.h "$work/include/"
rm -f "$work"/string_grouper_amd/csrc/*.o
make -s -j8 -C "$work/string_grouper_amd/csrc" \
  CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $*"
cp "$work/string_grouper_amd/libsg_hip.so" "$root/string_grouper_amd/libsg_hip_$name.so"
echo "built string_grouper_amd/libsg_hip_$name.so with: $*"
