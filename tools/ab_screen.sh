#!/bin/bash
# A/B of the decoder's screen on one box: ULTRA_HIP_LDPC_SCREEN=0 (off) against the default (on where the sample says it pays) —
# the decoder alone at converging / mixed / failing points of three codes, then the bench lines of cfg4, cfg5 and cfg3.
#   bash tools/ab_screen.sh > gpurun_out/ab_screen.txt
one() {
  label=$1; shift
  echo "== $label"
  env "$@" timeout -k 10 300 python3 - <<'PY' 2>/dev/null
import sys; sys.path.insert(0, ".")
import torch
from projectultra_amd import CodeRate, LDPCDecoder
for rate, es in ((0, 10.0), (0, 2.0), (0, 0.0), (0, -3.0), (4, 12.0), (4, 7.0), (4, 5.0), (4, 2.0), (2, 6.0), (5, 12.0)):
    ctx = LDPCDecoder(CodeRate(rate)).context
    llr, _ = ctx.make_llr_batch(1 << 17, es, seed=7)
    for _ in range(3): r = ctx.ldpc_decode(llr)
    ctx.synchronize(); ctx.timer_begin()
    for _ in range(10): r = ctx.ldpc_decode(llr)
    ms = ctx.timer_end() / 10
    it = r['iters'].float()
    print(f"rate {rate} Es/N0 {es:+.0f} dB: {ms:.4f} ms per 2^17 codewords, mean iterations {it.mean().item():.2f}, at once {(it == 0).float().mean().item():.3f}")
PY
  for cfg in cfg4 cfg5 cfg3; do
    env "$@" timeout -k 10 400 python3 bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-build 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('$cfg', d['value'], d['unit'], d['ms_per_step'], 'ms per step')"
  done
}
one "screen off" ULTRA_HIP_LDPC_SCREEN=0
one "screen on (default)" ULTRA_X=0
