#!/bin/bash
# GPU box: the other workloads' bench lines at HEAD (C3 / C4 / C5 with cpu_baseline + parity), the fp32-pinned training step, and
# two-rank rehearsals on the one GPU (gloo: mechanics, not measurements) -> gpurun_out/r03_extra/
O=gpurun_out/r03_extra; mkdir -p $O
for w in c3 c4 c5; do
  timeout -k 10 400 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$? $(cut -c1-200 $O/bench_$w.json)"
done
timeout -k 10 200 python bench.py --mode train --precision f32 > $O/bench_train_f32.json 2> $O/bench_train_f32.err; echo "train f32 rc=$? $(cut -c1-300 $O/bench_train_f32.json | grep -o '"ms_per_step": [0-9.]*')"
P=29540
for args in "--mode train" "--workload c4 --steps 2 --cpu-rays 0" "--steps 2 --cpu-rays 0 --no-c3-leg --no-fp32-leg --no-reference-order"; do
  P=$((P+1)); tag=$(echo $args | tr -c 'a-z0-9' '_' | cut -c1-24)
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 2 --rehearse-one-gpu $args > $O/rehearsal_2ranks_$tag.json 2> $O/rehearsal_2ranks_$tag.err
  echo "rehearsal [$args] rc=$? $(tail -1 $O/rehearsal_2ranks_$tag.json | cut -c1-160)"
done
