#!/bin/bash
# A/B of library variants on the decoder alone, same box: tools/ldpc_bench.py (2^17 mostly failing codewords per rate) and the
# R1/4 sweep's converged points (Es/N0 +10 dB) for the in-tree library and every build/v_*.so named in $VARIANTS.
#   VARIANTS="build/v_a.so build/v_b.so" bash tools/ab_ldpc.sh > gpurun_out/ab/x.txt
one() {
  label=$1; shift
  echo "== $label"
  env "$@" timeout -k 10 200 python3 tools/ldpc_bench.py 2>/dev/null
  env "$@" timeout -k 10 200 python3 - <<'PY' 2>/dev/null
import sys; sys.path.insert(0, ".")
import torch
from projectultra_amd import CodeRate, LDPCDecoder
for rate, es in ((0, 10.0), (0, -3.0), (4, 12.0), (2, 6.0)):
    ctx = LDPCDecoder(CodeRate(rate)).context
    llr, _ = ctx.make_llr_batch(1 << 17, es, seed=7)
    for _ in range(3): r = ctx.ldpc_decode(llr)
    ctx.synchronize(); ctx.timer_begin()
    for _ in range(10): r = ctx.ldpc_decode(llr)
    ms = ctx.timer_end() / 10
    print(f"rate {rate} Es/N0 {es:+.0f} dB: {ms:.4f} ms per 2^17 codewords, mean iterations {r['iters'].float().mean().item():.2f}")
PY
}
one in-tree ULTRA_X=0
for v in $VARIANTS; do one $v ULTRA_HIP_LIB=$v; done
