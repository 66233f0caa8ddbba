#!/bin/bash
# GPU box: PMC pass over the training step's kernels (grouped super-batch launches), one table line per kernel
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_r03_train_pmc; rm -rf $O; mkdir -p $O
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O -- python3 $PWD/bench.py --mode train --steps 3 --warmup 2 > $O.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not any(t in k for t in ("pny_mlp", "dw_gemm", "latent_grad", "pixel_linear")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
print("| kernel | dispatches | GRBM_GUI_ACTIVE | SQ_BUSY_CU | MFMA_BUSY | WAIT_INST_ANY | WAVE_CYCLES | matrix-busy | CU-busy |\n|---|---|---|---|---|---|---|---|---|")
for k, c in acc.items():
    n = len(disp[k]); g = c["GRBM_GUI_ACTIVE"] / 8
    print("| `%s` | %d | %.3g | %.3g | %.3g | %.3g | %.3g | %.0f %% | %.0f %% |" % (k, n, c["GRBM_GUI_ACTIVE"]/n, c["SQ_BUSY_CU_CYCLES"]/n, c["SQ_VALU_MFMA_BUSY_CYCLES"]/n,
          c["SQ_WAIT_INST_ANY"]/n, c["SQ_WAVE_CYCLES"]/n, 100*c["SQ_VALU_MFMA_BUSY_CYCLES"]/(1024*g), 100*c["SQ_BUSY_CU_CYCLES"]/(256*g)))
PY
