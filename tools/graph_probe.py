#!/usr/bin/env python3
"""Does replaying the cfg3 step as a HIP graph shorten it?  One step = demodulate + decode + count on a side stream,
captured with torch.cuda.graph after a warm-up (all workspaces exist), replayed against the same calls issued directly:
python3 tools/graph_probe.py [n_frames]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import torch
    from projectultra_amd import CodeRate, Modulation, ReceiveContext, presets
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    null_stream = len(sys.argv) > 2 and sys.argv[2] == "null"      # the direct calls on torch's default stream instead (no graph)
    mc = presets.nvis_mode().with_mode(Modulation.QAM16, CodeRate.R3_4)
    mc.pilot_spacing = 4
    s = torch.cuda.current_stream() if null_stream else torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx = ReceiveContext(mc)
        g = ctx.geometry
        audio, payload = ctx.make_batch(n, seed=0x5EED, first_frame=0, channel="watterson", snr_db=30.0, delay_ms=0.5, doppler_hz=0.1)
        out = dict(bytes=torch.empty((n, g.decoded_bytes), dtype=torch.uint8, device="cuda"), iters=torch.empty(n, dtype=torch.int32, device="cuda"),
                   ok=torch.empty(n, dtype=torch.uint8, device="cuda"), llr=torch.empty((n, g.llrs_per_frame), dtype=torch.float32, device="cuda"))
        counters = torch.zeros(8, dtype=torch.int64, device="cuda")

        def step():
            counters.zero_()
            r = ctx.demod_decode(audio, out=out)
            ctx.count_errors(r, payload, counters)

        for _ in range(3):
            step()
        s.synchronize()
        want = counters.clone()

        def timed(fn, k=20):
            s.synchronize(); t0 = time.perf_counter()
            for _ in range(k):
                fn()
            s.synchronize()
            return (time.perf_counter() - t0) / k * 1e3

        direct = min(timed(step) for _ in range(3))
        five = min(timed(step, 5) for _ in range(3))
    if null_stream:
        print(f"{n} frames on the default stream: direct {direct:.3f} ms per step (20 steps), {five:.3f} (5 steps)")
        return
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        step()
    with torch.cuda.stream(s):
        graph.replay(); s.synchronize()
        assert torch.equal(counters, want), (counters, want)
        replay = min(timed(graph.replay) for _ in range(3))
    print(f"{n} frames: direct {direct:.3f} ms per step, graph replay {replay:.3f} ms per step")


if __name__ == "__main__":
    main()
