#!/bin/bash
# GPU box: the stash store with its slot offset in an SGPR soffset -- as the compiler schedules it (soff), and with ONE wait state
# behind every such store by hand (soffnop) -- on the gradient tests that caught the anomaly
for v in soff soffnop base; do
  echo "=== stash store form: $v"
  if [ "$v" = base ]; then unset PNYOLO_LIB; else export PNYOLO_LIB=$PWD/build_dbg/libpnyolo_$v.so; fi
  timeout -k 10 600 python -m pytest tests/test_gpu_backward.py -q -m gpu --no-header -p no:cacheprovider -s \
     -k "f16x2_training_forward_against_fp32 or (default_arithmetic and dw_f16x2)" 2>&1 | grep -E "passed|failed|relative L2|AssertionError|worst gradient" | cut -c1-300
done
