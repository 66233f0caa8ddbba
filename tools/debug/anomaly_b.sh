#!/bin/bash
# GPU box: (1) the SLP build on the whole f16x2 parity suite + bench; (2) the stash-store forms on the gradient test that caught the anomaly
echo "=== slp0: f16x2 parity tests"; PNYOLO_LIB=$PWD/build_dbg/libpnyolo_slp0.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu --no-header -p no:cacheprovider -k "f16x2" 2>&1 | tail -2
tools/debug/bench_variants.sh base slp0 base slp0
for v in soff soffnop base; do
  echo "=== stash store form: $v"
  if [ "$v" = base ]; then unset PNYOLO_LIB; else export PNYOLO_LIB=$PWD/build_dbg/libpnyolo_$v.so; fi
  timeout -k 10 600 python -m pytest tests/test_gpu_backward.py -q -m gpu --no-header -p no:cacheprovider -s \
     -k "f16x2_training_forward_against_fp32 or (default_arithmetic and dw_f16x2)" 2>&1 | grep -E "passed|failed|relative L2|AssertionError|worst gradient" | cut -c1-400
done
