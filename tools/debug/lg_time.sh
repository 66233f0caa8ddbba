#!/bin/bash
# GPU box: per-variant duration of the latent-gradient kernel inside bench.py --mode train --train-encoder
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset PNYOLO_LIB; else export PNYOLO_LIB=$PWD/build_dbg/libpnyolo_$v.so; fi
  rm -rf gpurun_out/prof_lgv; rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/prof_lgv -- python3 $PWD/bench.py --mode train --train-encoder --steps 3 > gpurun_out/prof_lgv.log 2>&1
  echo "$v: $(grep -h latent_grad gpurun_out/prof_lgv/*/*kernel_stats.csv | awk -F, '{print "calls "$(NF-6)" avg_ns "$(NF-4)}')  step: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_lgv.log)"
done
