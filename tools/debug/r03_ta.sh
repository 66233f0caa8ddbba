#!/bin/bash
# round 3 (GPU box): vector-memory-pipeline counters (TA / TCP / TD) of the product MLP kernel, separate --pmc passes
# (two counters of a block per pass: more and rocprofv3 refuses the set, then hangs until killed)
set -o pipefail
OUT=$PWD/gpurun_out/prof_r03_ta
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 2 --warmup 1 --cpu-rays 0 --no-reference-order --no-fp32-leg --no-c3-leg"
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1
p=0
for set in "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum"; do
  p=$((p+1))
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pass$p" -- $BENCH > "$OUT/pass$p.log" 2>&1 || { echo "pass $p failed"; tail -3 "$OUT/pass$p.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/pass*/")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pny_mlp_h2_kernel" in r["Kernel_Name"]:
                acc[(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
    per = collections.defaultdict(list)
    for (c, disp), v in acc.items():
        per[c].append(sum(v))
    for c, v in sorted(per.items()):
        big = [x for x in v if x > 0.3 * max(v)] if max(v) > 0 else v
        print("%-40s mean over %d full launches %.5g" % (c, len(big), sum(big) / max(len(big), 1)))
PY
