#!/bin/bash
# Diagnostic builds of the f16x2 kernel (csrc/mlp_h2.hip) beside the product library:
#   tools/h2_variant_build.sh <name> [-DPNY_H2_STAMP] [-DPNY_H2_WD=4] [-DPNY_H2_NOSCHED] [-DPNY_H2_NOPREFETCH]
#                                    [-DPNY_H2_PLAIN_SPLIT] [-DPNY_H2_EXP_FOOT]
# -> build_dbg/libpnyolo_<name>.so (git-ignored, travels with gpurun).  Use with PNYOLO_LIB=$PWD/build_dbg/libpnyolo_<name>.so,
# e.g. `tools/debug/bench_variants.sh base <name>`; -DPNY_H2_STAMP prints the per-phase shares of every launch on stderr.
# The other objects are taken from the product build (run `make -C pixel-nerf-yolo_amd/csrc` first).
set -e
name=$1; shift
cd "$(dirname "$0")/../pixel-nerf-yolo_amd/csrc"
mkdir -p ../../build_dbg
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -fno-slp-vectorize "$@" \
    -c mlp_h2.hip -o /tmp/mlp_h2_variant.o
objs=$(ls *.o | grep -v '^mlp_h2.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/mlp_h2_variant.o -o ../../build_dbg/libpnyolo_$name.so
echo "built build_dbg/libpnyolo_$name.so ($*)"
