#!/bin/bash
for v in soff_none soff_n0 soff_n3 soff_n7; do
  echo "=== stash store, SGPR soffset form, wait states behind the store: $v"
  PNYOLO_LIB=$PWD/build_dbg/libpnyolo_$v.so timeout -k 10 600 python -m pytest tests/test_gpu_backward.py -q -m gpu --no-header -p no:cacheprovider -s \
     -k "f16x2_training_forward_against_fp32 and dw_f16x2" 2>&1 | grep -E "passed|failed|relative L2" | cut -c1-260
done
