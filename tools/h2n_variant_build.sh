#!/bin/bash
# Diagnostic builds of the 8-wave 16x16x32 kernel (csrc/mlp_h2n.hip = mlp_h2w.hip with PNY_HW_NW=8): see tools/h2w_variant_build.sh
set -e
name=$1; shift
cd "$(dirname "$0")/../pixel-nerf-yolo_amd/csrc"
mkdir -p ../../build_dbg
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -fno-slp-vectorize "$@" \
    -c mlp_h2n.hip -o /tmp/mlp_h2n_variant.o
objs=$(ls *.o | grep -v '^mlp_h2n.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/mlp_h2n_variant.o -o ../../build_dbg/libpnyolo_$name.so
echo "built build_dbg/libpnyolo_$name.so ($*)"
