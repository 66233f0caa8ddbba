#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py.
# Usage: [BENCH_ARGS="--projection off"] tools/profile_gpu.sh <tag>     -> gpurun_out/prof_<tag>/...
set -o pipefail
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 2 --warmup 1 --cpu-rays 0 --no-reference-order --no-fp32-leg --no-c3-leg ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1 || { echo "pmc fetch failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1 || { echo "pmc write failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1 || { echo "pmc sq failed"; tail -5 "$OUT/pmc_sq.log"; }
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_lds" -- $BENCH > "$OUT/pmc_lds.log" 2>&1 || { echo "pmc lds failed"; tail -5 "$OUT/pmc_lds.log"; }
find "$OUT" -name "*.csv" | head -30
du -sh "$OUT"
