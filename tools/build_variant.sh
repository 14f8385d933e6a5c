#!/bin/bash
# A/B builds: tools/build_variant.sh NAME FILE "-DFLAG ..."  ->  bio_amd/lib/NAME.so
# (FILE = kernels | kernels_affine | kernels_long | kernels_fasta | api, recompiled with the flags; the other objects are
# the ones `make` built).  Run with BIOALIGN_LIB=$GRAFT_REPO_ROOT/bio_amd/lib/NAME.so.
set -e
cd "$(dirname "$0")/.."
name=$1; file=$2; flags=$3
mkdir -p build/ab
src=bio_amd/csrc/$file.hip; x=""
if [ "$file" = api ]; then src=bio_amd/csrc/api.cpp; x="-x hip"; fi
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags $x -c -o build/ab/$name.o $src
objs=""
for o in kernels kernels_affine kernels_long kernels_fasta kernels_wide kernels_bp api; do
  if [ "$o" = "$file" ]; then objs="$objs build/ab/$name.o"; else objs="$objs build/$o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o bio_amd/lib/$name.so $objs
echo built bio_amd/lib/$name.so
