#!/bin/bash
# Diagnostic build of ONE translation unit of the library with extra flags, linked with the product objects:
#   tools/variant_build.sh <name> <source.hip> [-D...]   -> build_dbg/libpnyolo_<name>.so   (use with PNYOLO_LIB=...)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../pixel-nerf-yolo_amd/csrc"
mkdir -p ../../build_dbg
slp=""; case "$src" in mlp_h2*|mlp_bwd_h2*) slp="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result $slp "$@" -c $src -o /tmp/variant_$name.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/variant_$name.o -o ../../build_dbg/libpnyolo_$name.so
echo "built build_dbg/libpnyolo_$name.so ($src $*)"
