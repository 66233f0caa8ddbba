#!/bin/bash
# Diagnostic builds of the wide f16x2 kernel (csrc/mlp_h2w.hip) beside the product library (see tools/h2_variant_build.sh):
#   tools/h2w_variant_build.sh <name> [-DPNY_H2_STAMP] ...   -> build_dbg/libpnyolo_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../pixel-nerf-yolo_amd/csrc"
mkdir -p ../../build_dbg
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -fno-slp-vectorize "$@" \
    -c mlp_h2w.hip -o /tmp/mlp_h2w_variant.o
objs=$(ls *.o | grep -v '^mlp_h2w.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/mlp_h2w_variant.o -o ../../build_dbg/libpnyolo_$name.so
echo "built build_dbg/libpnyolo_$name.so ($*)"
