#!/bin/bash
# GPU box: parity of the wide f16x2 kernel on the golden tests, then its bench line beside the 8-wave shape's
mkdir -p gpurun_out
{
PNYOLO_H2_WIDE=1 PNYOLO_H2_SPLIT=0 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x --no-header -k "f16x2" 2>&1 | tail -15
for v in 0 1 0 1; do
  PNYOLO_H2_WIDE=$v timeout -k 10 200 python bench.py --steps 3 --cpu-rays 0 --no-reference-order --no-fp32-leg --no-c3-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('wide=$v %8.0f rays/s  %7.3f ms/launch  frac %.3f' % (d['value'], r['avg_launch_ms'], r['frac']))"
done
} > gpurun_out/r03_wide.log 2>&1
tail -25 gpurun_out/r03_wide.log
