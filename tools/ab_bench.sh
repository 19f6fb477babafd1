#!/bin/bash
# A/B of library variants on the GPU box: every build/v_*.so (tools/build_variants.sh) and the in-tree library go through
#   python3 bench.py <bench args> --no-cpu-baseline --no-build      (default args: --config cfg3 --frames 262144)
# and one line per variant with the step time and the per-kernel mean launch times is printed (and kept in $OUT).
#   OUT=gpurun_out/ab/x.txt ENVS="ULTRA_HIP_FALLBACK_CHAIN=1" bash tools/ab_bench.sh [bench args]
OUT=${OUT:-gpurun_out/ab/ab_$(date +%H%M%S).txt}
mkdir -p "$(dirname "$OUT")"
ARGS=${@:---config cfg3 --frames 262144}
run() {  # label, env assignments...
  label=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py $ARGS --no-cpu-baseline --no-build 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%-28s step %.3f ms  %.2f M/s | ' % ('$label', d['ms_per_step'], d['value']/1e6) + '  '.join('%s %.4f x%g' % (n.replace('_kernel',''), v['avg_launch_ms'], v['launches_per_step']) for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step'])))
" | tee -a "$OUT"
}
python3 bench.py --config cfg3 --frames 1024 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1   # builds the checker libraries once
run in-tree ULTRA_X=0
for e in $ENVS; do run "in-tree $e" $e; done
for v in build/v_*.so; do if [ -f "$v" ]; then run "$v" ULTRA_HIP_LIB=$v; fi; done
