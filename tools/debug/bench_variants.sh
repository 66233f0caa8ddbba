#!/bin/bash
# usage (on the GPU box): tools/debug/bench_variants.sh name1 name2 ...   (libraries build_dbg/libpnyolo_<name>.so; "base" = product)
for v in "$@"; do
  if [ "$v" = base ]; then unset PNYOLO_LIB; else export PNYOLO_LIB=$PWD/build_dbg/libpnyolo_$v.so; fi
  timeout -k 10 200 python bench.py --steps 3 --cpu-rays 0 --no-reference-order --no-fp32-leg --no-c3-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-12s %8.0f rays/s  %7.3f ms/launch  frac %.3f' % ('$v', d['value'], r['avg_launch_ms'], r['frac']))"
done
