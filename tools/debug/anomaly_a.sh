#!/bin/bash
# GPU box: does the packed-f32 (SLP) build of mlp_h2.hip still corrupt the gather blend, and what cures it?
for v in slp0 slpwait slpsb slp0; do
  echo "=== $v"
  PNYOLO_LIB=$PWD/build_dbg/libpnyolo_$v.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu --no-header -p no:cacheprovider \
     -k "f16x2_stress_deterministic" 2>&1 | grep -E "passed|failed|AssertionError|assert " | head -5
done
