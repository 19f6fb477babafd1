#!/bin/bash
# Everything under profiles/<tag>_* in one go, on the GPU box (three or four gpurun calls: stages a, b, c, d):
#   bash tools/collect_round.sh <tag> <commit> a|b|c|d
# a: cfg3 at 2^20 (default) and 2^18 + SQ counters;  b: cfg2, cfg5;  c: cfg4, raw;  d: stall tables (needs build/stamps.so
# of the SAME sources: STAMP_FLAGS="" bash tools/build_variants.sh), parity soaks, the torchrun line, LDPC rates, sweeps.
# tools/install_profiles.sh <tag> then copies the results from gpurun_out/ into profiles/.
set -u
TAG=${1:-r03}; C=${2:-unknown}; STAGE=${3:-a}
O=gpurun_out/$TAG; SQ=gpurun_out/${TAG}sq
mkdir -p $O $SQ
sq() { bash tools/pmc_sweep.sh $SQ/$1 --config "${@:2}" --no-cpu-baseline > $O/sq_$1.log 2>&1; rm -rf $SQ/$1/p?; grep -c '==' $SQ/$1/summary.txt; }
case $STAGE in
a) bash tools/collect_profiles.sh $TAG cfg3 $C > $O/c_cfg3.log 2>&1; head -4 $O/c_cfg3.log
   bash tools/collect_profiles.sh $TAG cfg3 $C '--frames 262144' _2e18 > $O/c_cfg3_2e18.log 2>&1; head -4 $O/c_cfg3_2e18.log
   sq cfg3 cfg3 --frames 262144 --steps 2 --warmup 1 ;;
b) for c in cfg2 cfg5; do bash tools/collect_profiles.sh $TAG $c $C > $O/c_$c.log 2>&1; head -4 $O/c_$c.log; done
   sq cfg2 cfg2 --steps 2 --warmup 1; sq cfg5 cfg5 --steps 1 --warmup 1 ;;
c) for c in cfg4 raw; do bash tools/collect_profiles.sh $TAG $c $C > $O/c_$c.log 2>&1; head -4 $O/c_$c.log; done
   sq cfg4 cfg4 --steps 1 --warmup 1; sq raw raw --steps 1 --warmup 1 ;;
d) timeout -k 10 300 python3 tools/mix_fft_stalls.py > $O/mix_fft_stalls_two_wave.txt 2> $O/stalls_two.err
   timeout -k 10 300 python3 tools/mix_fft_stalls.py --norot > $O/mix_fft_stalls_norot.txt 2> $O/stalls_norot.err
   timeout -k 10 300 python3 tools/ldpc_stalls.py > $O/ldpc_stalls.txt 2> $O/ldpc_stalls.err      # needs build/v_ldstamps.so (build_variants.sh ldstamps="-DUH_LDPC_STAMPS")
   timeout -k 10 300 python3 tools/ldpc_stalls_rate.py --rate 0 --esn0 -11 -3 10 > $O/ldpc_stalls_r14.txt 2>> $O/ldpc_stalls.err
   (cd oracle/_ref/tools && for cfg in "1024 6 4 8 30" "512 2 2 8 20"; do echo "=== reference build (CPU), $(grep -m1 "model name" /proc/cpuinfo | cut -d: -f2)"; ./live_latency.ref $cfg 2>/dev/null; echo "=== drop-in build (MI355X)"; ./live_latency.hip $cfg 2>/dev/null; done) > $O/live_latency.txt 2>&1
   timeout -k 10 400 python3 bench.py --config raw --raw-channel watterson > $O/bench_raw_watterson.json 2> $O/bench_raw_watterson.err
   # the N > 1 path on the one card (gloo): strong-scaling cfg3 at 2 and 4 ranks next to 1
   for n in 1 2 4; do timeout -k 10 400 python3 bench.py --gpus $n --backend gloo --total-frames 131072 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('ranks %d on one card (gloo): %.3f ms per 2^17-frame step, counters %s, collective world %d' % (d['n_gpus'], d['ms_per_step'], d['counters'], d['collective']['world_size']))"; done > $O/multirank_one_card.txt; cat $O/multirank_one_card.txt
   timeout -k 10 900 python3 tools/soak_parity.py 16384 7 > $O/soak_parity.txt 2>&1; tail -n 2 $O/soak_parity.txt
   timeout -k 10 600 python3 tools/soak_sync.py 4096 128 3 > $O/soak_sync.txt 2>&1; tail -n 2 $O/soak_sync.txt
   timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
       bench.py --gpus 1 --steps 5 --warmup 2 > $O/bench_cfg3_torchrun1.json 2> $O/torchrun.err; head -c 200 $O/bench_cfg3_torchrun1.json; echo
   timeout -k 10 200 python3 tools/ldpc_bench.py > $O/ldpc_bench.txt 2>&1; tail -n 6 $O/ldpc_bench.txt
   # what a rank of the strong-scaling run sees: the default batch of 2^20 frames cut into 1, 2, 4, 8 shares
   for n in 1048576 524288 262144 131072; do timeout -k 10 300 python3 bench.py --frames $n --no-cpu-baseline --no-build 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('%8d frames per GPU per step: %.3f ms, %.2f M frames/s per GPU (%d timed steps after %d untimed)' % (d['config']['launch_units'], d['ms_per_step'], d['value']/1e6, d['steps'], d['warmup'] + d['untimed_priming_steps']))"; done > $O/batch_size_series.txt; cat $O/batch_size_series.txt
   for c in cfg4 cfg5; do timeout -k 10 600 python3 tools/sweep.py --config $c --out $O/sweep_$c.json > $O/sweep_$c.txt 2>&1; tail -n 1 $O/sweep_$c.txt; done ;;
# e: stage d without the stall tables (a round that did not touch the kernels keeps the previous round's)
e) (cd oracle/_ref/tools && for cfg in "1024 6 4 8 30" "512 2 2 8 20"; do echo "=== reference build (CPU), $(grep -m1 "model name" /proc/cpuinfo | cut -d: -f2)"; ./live_latency.ref $cfg 2>/dev/null; echo "=== drop-in build (MI355X)"; ./live_latency.hip $cfg 2>/dev/null; done) > $O/live_latency.txt 2>&1
   timeout -k 10 400 python3 bench.py --config raw --raw-channel watterson > $O/bench_raw_watterson.json 2> $O/bench_raw_watterson.err
   # the N > 1 path on the one card (gloo): strong-scaling cfg3 at 2 and 4 ranks next to 1
   for n in 1 2 4; do timeout -k 10 400 python3 bench.py --gpus $n --backend gloo --total-frames 131072 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('ranks %d on one card (gloo): %.3f ms per 2^17-frame step, counters %s, collective world %d' % (d['n_gpus'], d['ms_per_step'], d['counters'], d['collective']['world_size']))"; done > $O/multirank_one_card.txt; cat $O/multirank_one_card.txt
   timeout -k 10 900 python3 tools/soak_parity.py 16384 7 > $O/soak_parity.txt 2>&1; tail -n 2 $O/soak_parity.txt
   timeout -k 10 600 python3 tools/soak_sync.py 4096 128 3 > $O/soak_sync.txt 2>&1; tail -n 2 $O/soak_sync.txt
   timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
       bench.py --gpus 1 --steps 5 --warmup 2 > $O/bench_cfg3_torchrun1.json 2> $O/torchrun.err; head -c 200 $O/bench_cfg3_torchrun1.json; echo
   timeout -k 10 200 python3 tools/ldpc_bench.py > $O/ldpc_bench.txt 2>&1; tail -n 6 $O/ldpc_bench.txt
   # what a rank of the strong-scaling run sees: the default batch of 2^20 frames cut into 1, 2, 4, 8 shares
   for n in 1048576 524288 262144 131072; do timeout -k 10 300 python3 bench.py --frames $n --no-cpu-baseline --no-build 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('%8d frames per GPU per step: %.3f ms, %.2f M frames/s per GPU (%d timed steps after %d untimed)' % (d['config']['launch_units'], d['ms_per_step'], d['value']/1e6, d['steps'], d['warmup'] + d['untimed_priming_steps']))"; done > $O/batch_size_series.txt; cat $O/batch_size_series.txt
   for c in cfg4 cfg5; do timeout -k 10 600 python3 tools/sweep.py --config $c --out $O/sweep_$c.json > $O/sweep_$c.txt 2>&1; tail -n 1 $O/sweep_$c.txt; done ;;
esac
