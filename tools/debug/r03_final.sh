#!/bin/bash
# end-of-round verification on the GPU box: suite, smoke, bench lines -> gpurun_out/r03_final/
set -o pipefail
O=gpurun_out/r03_final; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --mode train > $O/bench_train.json 2> $O/bench_train.err; echo "train rc=$?"
timeout -k 10 200 python bench.py --mode train --train-encoder > $O/bench_train_encoder.json 2> $O/bench_train_encoder.err; echo "train-enc rc=$?"
cat $O/bench.json $O/bench_train.json $O/bench_train_encoder.json | cut -c1-400
