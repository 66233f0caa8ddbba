#!/bin/bash
# round-3 experiment 1 (GPU box): MFMA-shape and stagger timing experiments + in-kernel clock
mkdir -p gpurun_out
{
tools/debug/bench_variants.sh base mfma16 stag8k stag16k base mfma16 stag8k stag16k
echo "--- stamp build"
PNYOLO_LIB=$PWD/build_dbg/libpnyolo_stamp.so timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-rays 0 --no-reference-order --no-fp32-leg --no-c3-leg 2>&1 | grep -E "h2 stamp|metric" | tail -8
} > gpurun_out/r03_exp1.log 2>&1
tail -20 gpurun_out/r03_exp1.log
