#!/usr/bin/env python3
"""Do the path's kernels overlap when the batch is split over HIP streams?  (One GPU.)

The decoder and the transform are bound by vector issue, the carrier half / pilot half / record kernels by HBM and latency
(DESIGN.md 4): 3.3 ms of the 21.6 ms headline step are kernels that leave the vector units mostly idle.  Split the 2^20-frame
batch into S sub-batches, one ReceiveContext and one HIP stream each, all S chains issued back to back — the hardware is free
to run a memory-bound kernel of one sub-batch beside a compute-bound kernel of another.  Prints ms per 2^20 frames for
S = 1, 2, 4 (same frames, same results: the chains are independent).

    python3 tools/two_stream_probe.py [total_frames]
"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from projectultra_amd import CodeRate, Modulation, ReceiveContext, presets  # noqa: E402


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    mc = presets.nvis_mode().with_mode(Modulation.QAM16, CodeRate.R3_4)
    mc.pilot_spacing = 4
    base = None
    for S in (1, 2, 4, 1):
        streams = [torch.cuda.Stream() for _ in range(S)]
        ctxs, bufs = [], []
        n = total // S
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                c = ReceiveContext(mc)
                audio, payload = c.make_batch(n, seed=0x5EED, first_frame=i * n, channel="watterson", snr_db=30.0)
                g = c.geometry
                out = dict(bytes=torch.empty((n, g.decoded_bytes), dtype=torch.uint8, device="cuda"),
                           iters=torch.empty(n, dtype=torch.int32, device="cuda"), ok=torch.empty(n, dtype=torch.uint8, device="cuda"),
                           llr=torch.empty((n, g.llrs_per_frame), dtype=torch.float32, device="cuda"))
                counters = torch.zeros(8, dtype=torch.int64, device="cuda")
                ctxs.append(c); bufs.append((audio, payload, out, counters))
        torch.cuda.synchronize()

        def step():
            for c, st, (audio, payload, out, counters) in zip(ctxs, streams, bufs):
                with torch.cuda.stream(st):
                    counters.zero_()
                    r = c.demod_decode(audio, out=out)
                    c.count_errors(r, payload, counters)
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 10
        for _ in range(K):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / K * 1e3
        tot = sum(b[3] for b in bufs).cpu().tolist()
        if base is None:
            base = tot
        print(f"{S} stream(s) x {n} frames: {ms:.3f} ms per {total} frames = {total / ms / 1e3:.2f} M frames/s; counters {'equal' if tot == base else 'DIFFER'} {tot[:3]}", flush=True)
        del ctxs, bufs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
