"""OFDMDemodulator — host-side mirror of ultra::OFDMDemodulator (include/ultra/ofdm.hpp:58-127)
for the part of its surface that is the hot path: the post-sync symbol loop and the
presynced entry, batched on the GPU.

`process()` is the chunk-fed receive of the reference: Schmidl-Cox search (scope row f1, on the GPU:
ultra_hip_acquire_batch) -> SYNCED symbol loop.  `process_synced()` / `processPresynced()` enter past the
search with the timing and CFO a sync stage produced.
"""
from __future__ import annotations

import math

import numpy as np

from . import _lib
from .engine import ReceiveContext
from .types import Entry, LDPC_BLOCK_SIZE, ModemConfig


class OFDMDemodulator:
    def __init__(self, config: ModemConfig, n_data_symbols=None, device=None, max_iterations: int = 50):
        self.config = config
        self._device = device
        self._n_data_symbols = n_data_symbols
        self._max_iterations = max_iterations
        self._ctx = {}                       # Entry -> ReceiveContext, built lazily
        self.reset()

    # -- reference API ---------------------------------------------------
    def reset(self) -> None:                 # demodulator.cpp:987-1016
        self._soft_bits = np.zeros(0, np.float32)
        self._cfo_hz = 0.0
        self._cfo_phase = 0.0
        self._chirp_cfo = False
        self._synced = False
        self._state = None
        self._rx = np.zeros(0, np.float32)   # everything fed so far (the search restarts from a fresh state)
        self._chunk = None                   # samples per process() call
        self._last_sync_offset = 0
        self._data_start = None

    def setFrequencyOffset(self, cfo_hz: float) -> None:            # demodulator.cpp:805-814
        self._cfo_hz, self._cfo_phase, self._chirp_cfo = float(cfo_hz), 0.0, True

    def setFrequencyOffsetWithPhase(self, cfo_hz: float, initial_phase_rad: float) -> None:   # :816-825
        self._cfo_hz, self._cfo_phase, self._chirp_cfo = float(cfo_hz), float(initial_phase_rad), True

    def getFrequencyOffset(self) -> float:
        return float(self._state[_lib_state("FREQ")]) if self._state is not None else self._cfo_hz

    def getEstimatedSNR(self) -> float:      # 10*log10(estimated_snr_linear), demodulator.cpp:797-799
        lin = float(self._state[_lib_state("SNR")]) if self._state is not None else 1.0
        return 10.0 * math.log10(lin) if lin > 0 else float("-inf")

    def isSynced(self) -> bool:
        return self._synced

    def hasPendingData(self) -> bool:
        return self._synced and self._soft_bits.size > 0

    def getSoftBits(self) -> np.ndarray:     # hands out 648 at a time (demodulator.cpp:766-791)
        if self._soft_bits.size <= LDPC_BLOCK_SIZE:
            out, self._soft_bits = self._soft_bits, np.zeros(0, np.float32)
            return out
        out, self._soft_bits = self._soft_bits[:LDPC_BLOCK_SIZE], self._soft_bits[LDPC_BLOCK_SIZE:]
        return out

    def processPresynced(self, samples, training_symbols: int = 2) -> bool:   # demodulator.cpp:854-985
        samples = np.ascontiguousarray(samples, np.float32).reshape(-1)
        sym = self.config.getSymbolDuration()
        if samples.size < sym:
            return False
        n_sym = samples.size // sym - training_symbols
        if n_sym <= 0:
            self._soft_bits = np.zeros(0, np.float32)
            return False
        ctx = self._context(Entry.PRESYNCED, n_sym, training_symbols)
        # a frequency offset that was never set travels as NaN: the library then estimates it from the two training
        # symbols (Impl::estimateCFOFromTraining, ofdm_sync.cpp:278-380; demodulator.cpp:920-925)
        return self._run(ctx, samples, never_set=not self._chirp_cfo)

    def process_synced(self, samples, cfo_hz: float = 0.0) -> bool:
        """SYNCED-state symbol loop of process() (demodulator.cpp:672-697) on a frame that
        starts at its first data symbol; cfo_hz = the coarse CFO the search stage sets."""
        samples = np.ascontiguousarray(samples, np.float32).reshape(-1)
        n_sym = samples.size // self.config.getSymbolDuration()
        if n_sym <= 0:
            return False
        self._cfo_hz, self._cfo_phase = float(cfo_hz), 0.0
        ctx = self._context(Entry.SYNCED, n_sym, 0)
        return self._run(ctx, samples)

    def process(self, samples) -> bool:
        """OFDMDemodulator::process (demodulator.cpp:461-700) for a stream fed in equal-sized calls (the
        harnesses feed 960 samples; the search result depends on the chunking): SEARCHING
        state on the GPU (ultra_hip_acquire_batch over everything fed so far), then the SYNCED symbol loop
        once the frame's samples have arrived.  True when at least 648 soft bits are buffered."""
        samples = np.ascontiguousarray(samples, np.float32).reshape(-1)
        if self._chunk is None:
            self._chunk = samples.size
        elif samples.size > self._chunk:
            raise ValueError("process(): calls must not grow (the chunk-fed search is emulated with a fixed call size)")
        self._rx = np.concatenate([self._rx, samples])
        ctx = self._context(Entry.SYNCED, None, 0)
        if not self._synced:
            if self._data_start is None:
                r = ctx.acquire(self._rx.reshape(1, -1), self._chunk)
                ctx.synchronize()
                if not int(r["found"][0]):
                    return False
                self._data_start = int(r["data_start"][0])
                self._cfo_hz, self._cfo_phase = float(r["cfo_hz"][0]), 0.0
                self._last_sync_offset = int(r["sync_offset"][0])
            fs = ctx.geometry.frame_samples
            if self._rx.size < self._data_start + fs:
                return False                 # the reference would have demodulated the symbols that are complete
            return self._run(ctx, self._rx[self._data_start:])
        return self._soft_bits.size >= LDPC_BLOCK_SIZE

    def getLastSyncOffset(self) -> int:      # demodulator.cpp:846-848
        return self._last_sync_offset

    # -- batch API --------------------------------------------------------
    def context(self, entry: Entry = Entry.SYNCED, n_data_symbols=None, training_symbols: int = 2) -> ReceiveContext:
        return self._context(Entry(entry), n_data_symbols, training_symbols)

    # ---------------------------------------------------------------------
    def _context(self, entry, n_sym, training):
        n_sym = n_sym if n_sym is not None else self._n_data_symbols
        key = (entry, n_sym, training)
        if key not in self._ctx:
            self._ctx[key] = ReceiveContext(self.config, entry=entry, n_data_symbols=n_sym,
                                            training_symbols=training, max_iterations=self._max_iterations,
                                            device=self._device)
        return self._ctx[key]

    def _run(self, ctx, samples, never_set: bool = False) -> bool:
        fs = ctx.geometry.frame_samples
        llr, state = ctx.demod(samples[:fs].reshape(1, fs), cfo_hz=np.array([np.nan if never_set else self._cfo_hz], np.float32),
                               cfo_phase=np.array([self._cfo_phase], np.float32), want_state=True)
        ctx.synchronize()
        self._soft_bits = np.concatenate([self._soft_bits, llr[0].cpu().numpy()])
        self._state = state[0].cpu().numpy()
        self._synced = True
        return self._soft_bits.size >= LDPC_BLOCK_SIZE


def _lib_state(name: str) -> int:
    return {"FREQ": 0, "NOISE": 1, "SNR": 2, "TIMING": 3, "PHASE": 4, "MIXER": 5, "SYMBOLS": 6}[name]
