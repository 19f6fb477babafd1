#!/bin/bash
# gpurun_out/<tag>* (tools/collect_round.sh) -> profiles/<tag>_*, then the issue model from the fresh counters.
set -u
TAG=${1:-r03}; O=gpurun_out/$TAG; SQ=gpurun_out/${TAG}sq; P=profiles
for c in cfg2 cfg3 cfg4 cfg5 raw; do
  cp $O/bench_$c.json $P/${TAG}_bench_$c.json; cp $O/kernel_stats_$c.csv $P/${TAG}_kernel_stats_$c.csv; cp $O/traffic_$c.json $P/${TAG}_traffic_$c.json
done
for f in bench kernel_stats traffic; do e=json; [ $f = kernel_stats ] && e=csv; cp $O/${f}_cfg3_2e18.$e $P/${TAG}_${f}_cfg3_2e18.$e; done
cp $O/bench_cfg3.json $P/${TAG}_bench_cfg3_2e20.json
cp $O/traffic_cfg3.json $P/traffic.json
cp $O/bench_cfg3_torchrun1.json $P/${TAG}_bench_cfg3_torchrun1.json
cp $SQ/cfg3/summary.txt $P/${TAG}_sq_counters.txt
for c in cfg2 cfg4 cfg5 raw; do cp $SQ/$c/summary.txt $P/${TAG}_sq_counters_$c.txt; done
cp $O/mix_fft_stalls_two_wave.txt $P/${TAG}_mix_fft_stalls_rot.txt; cp $O/mix_fft_stalls_norot.txt $P/${TAG}_mix_fft_stalls_norot.txt
cp $O/ldpc_stalls.txt $P/${TAG}_ldpc_stalls.txt; cp $O/bench_raw_watterson.json $P/${TAG}_bench_raw_watterson.json; cp $O/multirank_one_card.txt $P/${TAG}_multirank_one_card.txt
cat $O/soak_parity.txt $O/soak_sync.txt > $P/${TAG}_soak_parity.txt
cp $O/ldpc_bench.txt $P/${TAG}_ldpc_bench.txt
cp $O/batch_size_series.txt $P/${TAG}_batch_size_series.txt
for c in cfg4 cfg5; do cp $O/sweep_$c.json $P/${TAG}_sweep_$c.json; cp $O/sweep_$c.txt $P/${TAG}_sweep_$c.txt; done
python3 tools/issue_model.py --pmc $P/${TAG}_sq_counters.txt $P/${TAG}_sq_counters_cfg2.txt $P/${TAG}_sq_counters_cfg4.txt $P/${TAG}_sq_counters_cfg5.txt $P/${TAG}_sq_counters_raw.txt $P/r02_sq_counters_chirp.txt > $P/${TAG}_issue_model.txt
# the guarded compute roofline bench.py quotes (hash of the kernel sources inside): per-class issue cycles per work item
python3 tools/issue_model.py --json $P/issue.json --commit ${2:-unknown} --json-from cfg3=$SQ/cfg3 cfg2=$SQ/cfg2 cfg4=$SQ/cfg4 cfg5=$SQ/cfg5 raw=$SQ/raw
cp $O/ldpc_stalls_r14.txt $P/${TAG}_ldpc_stalls_r14.txt 2>/dev/null
cp $O/live_latency.txt $P/${TAG}_live_latency.txt 2>/dev/null
grep -l csrc_sha $P/${TAG}_traffic_*.json $P/traffic.json | xargs grep -h '"csrc_sha"' | sort | uniq -c
