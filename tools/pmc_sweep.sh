#!/bin/bash
# Collect SQ counters per kernel in separate rocprofv3 passes (no trace domains other than --kernel-trace).
# Usage (on the GPU box, from the repo root): bash tools/pmc_sweep.sh <outdir> [bench args...]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
PROG=${PMC_PROG:-bench.py}
ARGS=${@:---steps 2 --warmup 1 --no-cpu-baseline}
export TMPDIR=/tmp
mkdir -p "$OUT"
# the checker libraries are built HERE, by an unprofiled process; every profiled run passes --no-build (the profiler's
# preloaded library initialises the GPU before main(), and such a process must never start a compiler)
python3 -c "import bench; bench.ensure_built(False)" || exit 1
case "$PROG" in bench.py) ARGS="$ARGS --no-build";; esac
PASSES=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
 "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_VALU_MFMA_BUSY_CYCLES"
 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INSTS_SALU"
)
i=0
for p in "${PASSES[@]}"; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $p -d "$OUT/p$i" -o run -- python3 $PROG $ARGS > "$OUT/p$i.log" 2>&1
  echo "pass $i rc=$?"
  i=$((i+1))
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
# keep only the summary + small csvs
find "$OUT" -name "*.csv" -size +20M -delete
cat "$OUT/summary.txt"
