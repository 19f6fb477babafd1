"""LDPC decode time on the headline batch (cfg3: 2^18 frames, Watterson 30 dB): demodulate once, then time the decoder alone.
python3 tools/ldpc_cfg3_bench.py"""
import sys
sys.path.insert(0, ".")
import torch
from projectultra_amd import CodeRate, Modulation, ReceiveContext, presets
mc = presets.nvis_mode().with_mode(Modulation.QAM16, CodeRate.R3_4); mc.pilot_spacing = 4
ctx = ReceiveContext(mc)
n = 1 << 18
audio, payload = ctx.make_batch(n, seed=0x5EED, channel="watterson", snr_db=30.0, delay_ms=0.5, doppler_hz=0.1)
llr = ctx.demod(audio)[:, :648].contiguous()
for _ in range(2): r = ctx.ldpc_decode(llr)
ctx.synchronize()
ts = []
for _ in range(7):
    ctx.timer_begin(); r = ctx.ldpc_decode(llr); ts.append(ctx.timer_end())
print(f"ldpc cfg3 batch: best {min(ts):.3f} ms, median {sorted(ts)[3]:.3f} ms, mean iters {r['iters'].float().mean().item():.2f}, ok {r['ok'].float().mean().item():.4f}")
