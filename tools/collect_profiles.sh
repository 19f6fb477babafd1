#!/bin/bash
# Round-end evidence (run on the GPU box from the repo root):  bash tools/collect_profiles.sh <tag>
#   gpurun_out/<tag>/bench.json         python3 bench.py (defaults, with the CPU baseline)
#   gpurun_out/<tag>/kernel_stats.csv   rocprofv3 --kernel-trace --stats of bench.py --steps 3 --warmup 1
#   gpurun_out/<tag>/traffic.json       HBM bytes per launch from separate --pmc FETCH_SIZE / WRITE_SIZE passes
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks" -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/ks.log" 2>&1; echo "stats rc=$?"
cp "$OUT/ks/run_kernel_stats.csv" "$OUT/kernel_stats.csv" 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $c -d "$OUT/pmc_$c" -o run -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_$c.log" 2>&1; echo "pmc $c rc=$?"
done
python3 tools/traffic_from_pmc.py "$OUT" > "$OUT/traffic.json"
rm -rf "$OUT/ks" "$OUT"/pmc_*/ 2>/dev/null
cat "$OUT/bench.json"; head -8 "$OUT/kernel_stats.csv" | cut -c1-160; cat "$OUT/traffic.json"
