#!/bin/bash
# GPU box: the native training trunk -- tests, step time beside the ATen graph, kernel trace
mkdir -p gpurun_out
{
timeout -k 10 600 python -m pytest tests/test_gpu_backward.py -q -m gpu --no-header -s -k "encoder_training_gradients or trunk_training_batch" 2>&1 | grep -E "passed|failed|worst relative|Error" | cut -c1-200
for t in native torch native torch; do
  PNYOLO_TRUNK=$t timeout -k 10 300 python bench.py --mode train --train-encoder --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train-encoder trunk=$t  %.2f ms/step (min/med/max %s)  loss %.4f -> %.4f' % (d['ms_per_step'], ' '.join('%.2f' % x for x in d['ms_per_step_min_median_max']), d['loss_first'], d['loss_last']))"
done
timeout -k 10 300 python bench.py --mode train --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('frozen trunk  %.2f ms/step' % d['ms_per_step'])"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_trunk -- python3 bench.py --mode train --train-encoder --steps 10 > gpurun_out/prof_r03_trunk.log 2>&1
f=$(find gpurun_out/prof_r03_trunk -name "*kernel_stats.csv" | head -1)
echo "kernel stats: $f"; head -40 "$f" | cut -c1-150
} > gpurun_out/r03_trunk.log 2>&1
tail -70 gpurun_out/r03_trunk.log
