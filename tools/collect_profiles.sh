#!/bin/bash
# Round-end evidence (run on the GPU box from the repo root):  bash tools/collect_profiles.sh <tag> [config] [commit]
#   gpurun_out/<tag>/bench_<config>.json        python3 bench.py --config <config> (defaults, with the CPU baseline)
#   gpurun_out/<tag>/kernel_stats_<config>.csv  rocprofv3 --kernel-trace --stats of bench.py --steps 3 --warmup 1
#   gpurun_out/<tag>/traffic_<config>.json      HBM bytes per launch from separate --pmc FETCH_SIZE / WRITE_SIZE passes
# The checker libraries are (re)built by the FIRST, unprofiled run; every run under rocprofv3 passes --no-build: the
# profiler's preloaded library initialises the GPU before main(), and such a process must never start a compiler.
set -u
TAG=${1:-r03}
CFG=${2:-cfg3}
COMMIT=${3:-unknown}
EXTRA=${4:-}            # extra bench.py arguments (e.g. "--frames 262144"), output files then carry the suffix $5
SUF=${5:-}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 900 python3 bench.py --config $CFG $EXTRA > "$OUT/bench_$CFG$SUF.json" 2> "$OUT/bench_$CFG.err"; echo "bench rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks" -o run -- python3 bench.py --config $CFG $EXTRA --steps 3 --warmup 1 --no-cpu-baseline --no-build > "$OUT/ks_$CFG.log" 2>&1; echo "stats rc=$?"
cp "$OUT/ks/run_kernel_stats.csv" "$OUT/kernel_stats_$CFG$SUF.csv" 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv --pmc $c -d "$OUT/pmc_$c" -o run -- python3 bench.py --config $CFG $EXTRA --steps 1 --warmup 0 --no-cpu-baseline --no-build > "$OUT/pmc_${c}_$CFG.log" 2>&1; echo "pmc $c rc=$?"
done
UNITS=$(python3 -c "import json;print(json.load(open('$OUT/bench_$CFG$SUF.json'))['config']['launch_units'])")
python3 tools/traffic_from_pmc.py "$OUT" $CFG $UNITS $COMMIT > "$OUT/traffic_$CFG$SUF.json"
rm -rf "$OUT/ks" "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE 2>/dev/null
head -c 600 "$OUT/bench_$CFG$SUF.json"; echo; head -12 "$OUT/kernel_stats_$CFG$SUF.csv" | cut -c1-170; cat "$OUT/traffic_$CFG$SUF.json"
