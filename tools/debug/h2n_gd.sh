#!/bin/bash
# GPU box: gather depth of the 8-wave 16x16x32 kernel (PNYOLO_H2_WIDE=2), libraries build_dbg/libpnyolo_<name>.so
run() {  # name wide
  if [ "$1" = base ]; then unset PNYOLO_LIB; else export PNYOLO_LIB=$PWD/build_dbg/libpnyolo_$1.so; fi
  PNYOLO_H2_WIDE=$2 timeout -k 10 200 python bench.py --steps 3 --cpu-rays 0 --no-reference-order --no-fp32-leg --no-c3-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-10s wide=%s %8.0f rays/s  %7.3f ms/launch  frac %.3f' % ('$1', '$2', d['value'], r['avg_launch_ms'], r['frac']))"
}
for rep in 1 2; do run base 0; run base 2; for v in "$@"; do run $v 2; done; done
