#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE (separate passes) of the training step's kernels
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  O=$PWD/gpurun_out/prof_r03_train_$c; rm -rf $O
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -- python3 $PWD/bench.py --mode train --steps 3 --warmup 2 > $O.log 2>&1
  python3 - $O $c <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not any(t in k for t in ("pny_mlp", "dw_gemm", "pixel_linear", "dw_reduce")): continue
        acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc: print("%-10s %-50s dispatches %3d  mean %.4g KB" % (sys.argv[2], k[:50], len(n[k]), acc[k] / len(n[k])))
PY
done
