"""Where a mixed decode launch's time goes: the codewords that converge against the codewords that never do.

VERDICT r5 item 7(b) asked for cfg5 >= 48 M frames/s "via the iterating kernel's ~9,000-cycle head on 1-2-iteration codewords".
This measures the ceiling of that lever directly.  For every code rate: a launch shaped like one of cfg5's six decoder launches —
105,600 codewords (5 modulations x 11 SNR points x 1,920 frames), BPSK/AWGN soft bits at 11 Es/N0 points one dB apart placed so that
about a third of the codewords never converge (cfg5: 32.8 %) — decoded (a) whole, (b) only the codewords of (a) that converged,
(c) only those that did not.  (b) is everything a faster head could ever touch.

Run on the GPU box:  python3 tools/decoder_time_split.py          (the synthetic ladder: more near-waterfall points than cfg5 has)
                     python3 tools/decoder_time_split.py --cfg5   (cfg5 ITSELF: the soft bits its own demodulations produce, rate by rate)"""
import sys

import torch

sys.path.insert(0, ".")
from projectultra_amd import CodeRate, LDPCDecoder  # noqa: E402

N_POINTS, PER_POINT = 11, 9600


def timed(ctx, llr, reps=5):
    for _ in range(2):
        r = ctx.ldpc_decode(llr)
    ctx.synchronize()
    ctx.timer_begin()
    for _ in range(reps):
        r = ctx.ldpc_decode(llr)
    return ctx.timer_end() / reps, r


def batch(ctx, lo_db):
    parts = [ctx.make_llr_batch(PER_POINT, lo_db + p, seed=0x5EED + p)[0] for p in range(N_POINTS)]
    return torch.cat(parts, dim=0).contiguous()


def cfg5():
    """The bench's cfg5 grid (5 modulations x 6 rates x 11 SNR points x 1,920 frames): demodulate once, then decode each rate's
    105,600 codewords whole / converging only / failing only, and ask the screen how many of the converging ones were clean."""
    from projectultra_amd.sweep import HipModeGrid
    grid = HipModeGrid()
    grid.generate(0)
    grid.receive()
    torch.cuda.synchronize()
    M, R, S, n = len(grid.mods), len(grid.rates), len(grid.snrs), grid.n
    rows = S * n
    tot = {"all": 0.0, "conv": 0.0, "fail": 0.0, "cw": 0, "conv_cw": 0, "dirty": 0}
    print(f"{'rate':6s} {'fail %':>7s} {'whole launch':>13s} {'converging only':>16s} {'failing only':>13s} {'converging: clean at receipt (screen) / iterated':>50s}")
    for ri, ctx in enumerate(grid.ldpc_ctx):
        llr = torch.cat([grid.llr[(mi * R + ri) * rows:(mi * R + ri + 1) * rows, :648] for mi in range(M)], dim=0).contiguous()
        t_all, r = timed(ctx, llr)
        ok = r["ok"].bool()
        conv, failing = llr[ok].contiguous(), llr[~ok].contiguous()
        ctx.clear_status()
        t_conv, _ = timed(ctx, conv)
        st = ctx.status()
        t_fail, _ = timed(ctx, failing)
        dirty = st["screen_dirty"] if st["screen_gate_open"] else conv.shape[0]
        print(f"R{['1/4', '1/3', '1/2', '2/3', '3/4', '5/6'][int(grid.rates[ri])]:5s} {100 * (1 - ok.float().mean().item()):6.1f}% {t_all:10.3f} ms {t_conv:13.3f} ms {t_fail:10.3f} ms "
              f"{conv.shape[0] - dirty:28d} / {dirty}")
        tot["all"] += t_all; tot["conv"] += t_conv; tot["fail"] += t_fail; tot["cw"] += llr.shape[0]; tot["conv_cw"] += conv.shape[0]; tot["dirty"] += dirty
    print(f"\ncfg5's six decode launches: whole {tot['all']:.3f} ms; the {tot['conv_cw']} converging codewords alone {tot['conv']:.3f} ms "
          f"({100 * tot['conv'] / tot['all']:.1f} %), of which {tot['conv_cw'] - tot['dirty']} are finished by the screen and {tot['dirty']} iterate; "
          f"the {tot['cw'] - tot['conv_cw']} failing ones alone {tot['fail']:.3f} ms")
    print("A head that cost NOTHING on the iterated converging codewords would save less than the middle column.")


if "--cfg5" in sys.argv:
    cfg5()
    sys.exit(0)

tot = {"all": 0.0, "conv": 0.0, "fail": 0.0}
print(f"{'rate':6s} {'fail %':>7s} {'whole launch':>13s} {'converging only':>16s} {'failing only':>13s} {'mean iters (conv)':>18s}   iterations on failing codewords")
for rate in range(6):
    ctx = LDPCDecoder(CodeRate(rate)).context
    # place the 11-point ladder so that ~1/3 of the codewords fail: walk its lowest point up from -14 dB
    lo = -14.0
    while True:
        llr = batch(ctx, lo)
        r = ctx.ldpc_decode(llr)
        fail = 1.0 - r["ok"].float().mean().item()
        if fail <= 0.36 or lo > 12.0:
            break
        lo += 0.5
    t_all, r = timed(ctx, llr)
    ok = r["ok"].bool()
    conv, failing = llr[ok].contiguous(), llr[~ok].contiguous()
    t_conv, rc = timed(ctx, conv)
    t_fail, _ = timed(ctx, failing)
    it_conv = rc["iters"].float().mean().item()
    iters_fail = 50 * int((~ok).sum())
    iters_all = int(r["iters"].sum()) + int(ok.sum())          # executed iterations: index + 1 for those that stopped, 50 for the rest
    print(f"R{['1/4', '1/3', '1/2', '2/3', '3/4', '5/6'][rate]:5s} {100 * fail:6.1f}% {t_all:10.3f} ms {t_conv:13.3f} ms {t_fail:10.3f} ms {it_conv:18.2f}   "
          f"{100 * iters_fail / (iters_fail + int(rc['iters'].sum()) + int(ok.sum())):.1f} % of the launch's executed iterations")
    tot["all"] += t_all; tot["conv"] += t_conv; tot["fail"] += t_fail
print(f"\nsix launches (one cfg5 step's decoding): whole {tot['all']:.3f} ms; the converging codewords alone {tot['conv']:.3f} ms "
      f"({100 * tot['conv'] / tot['all']:.1f} %); the failing ones alone {tot['fail']:.3f} ms")
print("A head that cost NOTHING on converging codewords would save at most the middle column: the ceiling of VERDICT r5 7(b)'s lever.")
