#!/bin/bash
# GPU box: parity of the 8-wave 16x16x32 shape (PNYOLO_H2_WIDE=2), then bench beside the product kernel and the 4-wave shape
{
PNYOLO_H2_WIDE=2 PNYOLO_H2_SPLIT=0 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu --no-header -k "f16x2 and not split_shape" 2>&1 | tail -3
for v in 0 2 1 0 2; do
  PNYOLO_H2_WIDE=$v timeout -k 10 200 python bench.py --steps 3 --cpu-rays 0 --no-reference-order --no-fp32-leg --no-c3-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('wide=$v %8.0f rays/s  %7.3f ms/launch  frac %.3f' % (d['value'], r['avg_launch_ms'], r['frac']))"
done
} 2>&1 | tee gpurun_out/r03_h2n.log
