"""Per-kernel times of one decode launch with the screen on: run under rocprofv3 --kernel-trace --stats.
   python3 tools/screen_probe.py <rate> <esn0_db> [n_cw]"""
import sys; sys.path.insert(0, ".")
import torch
from projectultra_amd import CodeRate, LDPCDecoder
rate, es = int(sys.argv[1]), float(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 17
ctx = LDPCDecoder(CodeRate(rate)).context
llr, _ = ctx.make_llr_batch(n, es, seed=7)
for _ in range(12): r = ctx.ldpc_decode(llr)
ctx.synchronize()
it = r['iters'].float()
print(f"rate {rate} Es/N0 {es:+.1f} dB: mean iterations {it.mean().item():.2f}, at once {(it == 0).float().mean().item():.3f}")
