#!/bin/bash
# GPU box: how wide a relu margin do the comparisons of the DEFAULT training arithmetic (f16x2 projected forward, f16x2
# backward) with autograd through the oracle need?  Runs the two default-arithmetic tests of tests/test_gpu_backward.py at
# several margins and collects the worst gradient error (or the failing tensor) of each -> gpurun_out/r03_margin_sweep.log
out=gpurun_out/r03_margin_sweep.log
mkdir -p gpurun_out; : > $out
for m in 0 1e-5 2e-5 3e-5 5e-5 1e-4; do
  echo "=== margin $m" >> $out
  PNYOLO_TEST_AMBIG_DEFAULT=$m timeout -k 10 600 python -m pytest tests/test_gpu_backward.py -q -m gpu -s -x --no-header -p no:cacheprovider \
      -k "default_arithmetic and dw_f16x2" 2>&1 | grep -E "default arithmetic|AssertionError|passed|failed|not enough" >> $out
done
cat $out
