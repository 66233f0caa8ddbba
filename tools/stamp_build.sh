#!/bin/bash
# Diagnostic build: libpnyolo_stamp.so with s_memtime phase stamps in the MLP kernel (-DPNY_STAMP).
# Use with PNYOLO_LIB=$PWD/pixel-nerf-yolo_amd/libpnyolo_stamp.so python bench.py --steps 1 --cpu-rays 0
set -e
cd "$(dirname "$0")/../pixel-nerf-yolo_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DPNY_STAMP"
/opt/rocm/bin/hipcc $FLAGS -shared api.hip mlp.hip render_kernels.hip encoder.hip detect.hip -o ../libpnyolo_stamp.so
echo built ../libpnyolo_stamp.so
