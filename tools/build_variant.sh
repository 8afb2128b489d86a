#!/bin/bash
# Builds a copy of the library from a directory of replacement sources:
#   tools/build_variant.sh NAME [FILE...]   (files replace their namesakes from renderer_amd/csrc)
# -> build/lib_NAME.so, selected at run time with RENDERER_AMD_LIB.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/rnd_variant_$NAME
rm -rf $W; mkdir -p $W/renderer_amd/csrc $ROOT/build
ln -s $ROOT/include $W/include
cp $ROOT/renderer_amd/csrc/*.h $ROOT/renderer_amd/csrc/*.hip $ROOT/renderer_amd/csrc/*.cpp $W/renderer_amd/csrc/
for f in "$@"; do cp "$f" $W/renderer_amd/csrc/$(basename "$f" | sed 's/^[^.]*__//'); done
cd $W/renderer_amd/csrc
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 ${TRACKERS--mllvm -amdgpu-use-amdgpu-trackers=1} $EXTRA -shared -o $ROOT/build/lib_$NAME.so rnd_kernels.hip rnd_abi.cpp
