"""CPU stand-ins for the per-rank GPU shards of projectultra_amd.sweep (TEST INFRASTRUCTURE): the same
`run(lo, hi, snr_db, seed) -> int64[8]` interface, computed by the oracle on the host.  The sweep driver's loops
(sharding, per-point seeds, the one all-reduce per point, curve assembly) are the product and run unchanged on top.

OracleLdpcShard is more than a stand-in: the device's LLR generator is bit-identical to the oracle's twin, so its
counters are exactly what HipLdpcShard must report for the same (seed, index range, Es/N0)."""
import numpy as np
import torch

from oracle.bindings import INFO_BITS, geometry, oracle as get_oracle


def count(bytes_, iters, ok, payload):
    """Host recount of the eight Monte-Carlo counters (ultra_hip_count_errors semantics)."""
    pb = payload.shape[1]
    diff = np.unpackbits(bytes_[:, :pb] ^ payload, axis=1).sum(axis=1)
    okb = ok.astype(bool)
    return np.array([len(okb), int(((~okb) | (diff > 0)).sum()), int(diff.sum()), 8 * pb * len(okb), int((~okb).sum()),
                     int(iters.sum()), int((okb & (diff > 0)).sum()), 0], np.int64)


class OracleLdpcShard:
    def __init__(self, rate, max_iterations=50, n_threads=8):
        self.rate, self.max_iterations, self.n_threads = int(rate), max_iterations, n_threads
        self.o = get_oracle()

    def run(self, lo, hi, snr_db, seed):
        if hi <= lo:
            return torch.zeros(8, dtype=torch.int64)
        llr, payload = self.o.make_llr_batch(self.rate, hi - lo, snr_db, seed=seed, c0=lo)
        by, it, ok = self.o.ldpc_decode_batch_mt(self.rate, llr, self.n_threads, self.max_iterations)
        return torch.from_numpy(count(by, it, ok, payload))


class OracleModemShard:
    def __init__(self, cfg, channel="awgn", n_threads=8):
        self.cfg, self.channel, self.n_threads = cfg, channel, n_threads
        self.o = get_oracle()

    def run(self, lo, hi, snr_db, seed):
        if hi <= lo:
            return torch.zeros(8, dtype=torch.int64)
        audio, payload = self.o.make_batch(self.cfg, hi - lo, seed=seed, f0=lo, channel=self.channel, snr_db=snr_db,
                                           n_threads=self.n_threads)
        r = self.o.demod_decode_batch(self.cfg, audio, n_threads=self.n_threads, want_llr=False, want_state=False)
        return torch.from_numpy(count(r["bytes"], r["iters"], r["ok"], payload))

    def run_points(self, lo, hi, snr_points, seeds):
        """HipModemShard.run_points' contract: row i = run(lo, hi, snr_points[i], seeds[i]); the driver reduces the block."""
        return torch.stack([self.run(lo, hi, s, sd) for s, sd in zip(snr_points, seeds)]) if len(snr_points) else torch.zeros((0, 8), dtype=torch.int64)
