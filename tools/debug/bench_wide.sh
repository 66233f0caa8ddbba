#!/bin/bash
# usage (GPU box): tools/debug/bench_wide.sh name1 name2 ...  ("base" = product library; every run with PNYOLO_H2_WIDE=1 unless the name is "narrow")
for v in "$@"; do
  export PNYOLO_H2_WIDE=1
  if [ "$v" = base ]; then unset PNYOLO_LIB; elif [ "$v" = narrow ]; then unset PNYOLO_LIB; export PNYOLO_H2_WIDE=0; else export PNYOLO_LIB=$PWD/build_dbg/libpnyolo_$v.so; fi
  timeout -k 10 200 python bench.py --steps 3 --cpu-rays 0 --no-reference-order --no-fp32-leg --no-c3-leg 2>/tmp/bw_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-12s %8.0f rays/s  %7.3f ms/launch  frac %.3f' % ('$v', d['value'], r['avg_launch_ms'], r['frac']))"
  grep "stamp" /tmp/bw_err.log | tail -2
done
