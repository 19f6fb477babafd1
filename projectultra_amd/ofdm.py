"""OFDMDemodulator — host-side mirror of ultra::OFDMDemodulator (include/ultra/ofdm.hpp:58-127)
for the part of its surface that is the hot path: the post-sync symbol loop and the
presynced entry, batched on the GPU.

`process()` is the chunk-fed receive of the reference as a live stream: incremental Schmidl-Cox search (scope row f1:
ultra_hip_acquire_stream_batch) -> SYNCED symbol loop as the symbols arrive (ultra_hip_demod_stream_batch) -> the exits of SYNCED.  `process_synced()` / `processPresynced()` enter past the
search with the timing and CFO a sync stage produced.
"""
from __future__ import annotations

import math

import numpy as np

from . import _lib
from .engine import ReceiveContext
from .types import Entry, LDPC_BLOCK_SIZE, ModemConfig


class OFDMDemodulator:
    MAX_CONSTELLATION_SYMBOLS = 500          # demodulator_constants.hpp:122

    def __init__(self, config: ModemConfig, n_data_symbols=None, device=None, max_iterations: int = 50):
        self.config = config
        self._constellation = np.zeros(0, np.complex64)    # never cleared, as in the reference (not even by reset())
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
        self._resume = None                  # live-stream state of process(): built on first use
        self._synced_symbols = self._idle_calls = 0
        self._origin = self._fed = 0

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

    MAX_SYMBOLS_BEFORE_TIMEOUT, MAX_IDLE_CALLS_BEFORE_RESET = 250, 10       # demodulator_constants.hpp:37-38

    def process(self, samples) -> bool:
        """OFDMDemodulator::process (demodulator.cpp:461-741) on a LIVE stream, call by call — the Python twin of
        ultra_hip::HipOfdmCoxWaveform: while SEARCHING one launch of the chunk-fed Schmidl-Cox search per call, resumed
        from the previous call's state (ultra_hip_acquire_stream_batch); once SYNCED whole symbols are demodulated as
        they arrive, the tracker continuing on the device (ultra_hip_demod_stream_batch); SYNCED is left after more than
        250 symbols, more than 10 calls without a new soft bit, or an empty call with nothing left (frame complete), and
        the search restarts on what is still buffered; a new preamble arriving while SYNCED (symbols demodulated, two calls
        or more without a soft bit, six preamble symbols buffered: :605-657) restarts the demodulation on the new frame
        (ultra_hip_resync_stream_batch).  True when at least 648 soft bits are buffered."""
        import torch
        samples = np.ascontiguousarray(samples, np.float32).reshape(-1)
        ctx = self._context(Entry.SYNCED, self.MAX_SYMBOLS_BEFORE_TIMEOUT + 1, 0)
        if getattr(self, "_resume", None) is None:
            self._origin = self._fed = 0
            self._resume = torch.zeros((1, 4), dtype=torch.int32, device=ctx.device)
            self._synced_symbols = self._idle_calls = 0
        self._rx = np.concatenate([self._rx, samples])
        self._fed += samples.size
        # Absolute sample indices travel to the device as 32-bit words (and ultra_hip_acquire_stream_batch refuses more than
        # 2^30 - 1 samples): a long-lived stream rebases them on the start of its buffer, as HipOfdmCoxWaveform::rebase() does
        # — while SEARCHING, where nothing but the resume words refers to them.
        if not self._synced and self._origin > 0 and self._fed > (1 << 29):
            shift = self._origin
            words = self._resume[0].cpu().tolist()
            self._resume.copy_(torch.tensor([[_i32((words[0] & 0xffffffff) - shift), _i32((words[1] & 0xffffffff) - shift), words[2], words[3]]],
                                            dtype=torch.int32))
            self._fed -= shift
            self._origin = 0
        if not self._synced:
            window = torch.from_numpy(self._rx if self._rx.size else np.zeros(1, np.float32)).reshape(1, -1)
            r = ctx.acquire_stream(window, self._origin, self._fed, self._resume)
            ctx.synchronize()
            base = int(self._resume[0, 0].item()) & 0xffffffff
            if int(r["found"][0]):
                self._cfo_hz, self._cfo_phase = float(r["cfo_hz"][0]), 0.0
                self._coarse_cfo = self._cfo_hz
                self._last_sync_offset = int(r["sync_offset"][0])
                self._consume_to(int(r["data_start"][0]))
                self._synced, self._synced_symbols, self._state = True, 0, None
            elif base > self._origin:
                self._consume_to(base)                    # what the search trimmed off rx_buffer
        if not self._synced:
            return False
        sym = ctx.geometry.symbol_samples
        preamble_total = 6 * (self.config.fft_size + ctx.geometry.cp_len)
        if self._synced_symbols > 0 and self._idle_calls >= 2 and self._rx.size >= preamble_total:       # :605-657
            here = torch.tensor([[_i32(self._origin), _i32(self._fed), 0, 0]], dtype=torch.int32, device=ctx.device)
            r = ctx.acquire_stream(torch.from_numpy(self._rx).reshape(1, -1), self._origin, self._fed, here, midframe=True)
            ctx.synchronize()
            if int(r["found"][0]):
                self._cfo_hz, self._cfo_phase = float(r["cfo_hz"][0]), 0.0
                self._coarse_cfo = self._cfo_hz
                self._consume_to(int(r["data_start"][0]))
                self._soft_bits = np.zeros(0, np.float32)
                self._synced_symbols, self._idle_calls, self._state = 0, 0, None
        n_new = min(self._rx.size // sym, self.MAX_SYMBOLS_BEFORE_TIMEOUT + 1 - self._synced_symbols)
        before = self._soft_bits.size
        if n_new > 0:
            llr, state, eq = ctx.demod_stream(self._rx[:n_new * sym].reshape(1, -1), self._synced_symbols, n_new,
                                              cfo_hz=np.array([self._cfo_hz], np.float32) if self._synced_symbols == 0 else None,
                                              want_state=True, want_equalized=True)
            ctx.synchronize()
            self._soft_bits = np.concatenate([self._soft_bits, llr[0].cpu().numpy()])
            # demodulateSymbol (demodulator.cpp:199-208): every data symbol appends its equalized carriers, the newest 500 stay
            self._constellation = np.concatenate([self._constellation, eq[0].cpu().numpy().reshape(-1)])[-self.MAX_CONSTELLATION_SYMBOLS:]
            self._state = state[0].cpu().numpy()
            self._consume_to(self._origin + n_new * sym)
            self._synced_symbols += n_new
            if self._synced_symbols > self.MAX_SYMBOLS_BEFORE_TIMEOUT:            # :683-691
                self._to_searching()
                return self._soft_bits.size >= LDPC_BLOCK_SIZE
        if self._soft_bits.size == before:                                        # :704-716
            self._idle_calls += 1
            if self._idle_calls > self.MAX_IDLE_CALLS_BEFORE_RESET:
                self._to_searching()
                return self._soft_bits.size >= LDPC_BLOCK_SIZE
        else:
            self._idle_calls = 0
        has_codeword = self._soft_bits.size >= LDPC_BLOCK_SIZE
        if not has_codeword and self._synced_symbols > 0 and samples.size == 0 and n_new == 0:   # frame complete (:720-731)
            self._to_searching()
            self._soft_bits = np.zeros(0, np.float32)
        return has_codeword

    def _consume_to(self, abs_index: int) -> None:        # rx_buffer.erase(begin, begin + n)
        self._rx = self._rx[abs_index - self._origin:]
        self._origin = abs_index

    def _to_searching(self) -> None:
        """Back to SEARCHING on whatever is still buffered; the energy gate's noise floor survives (Impl member)."""
        import torch
        self._synced, self._synced_symbols, self._idle_calls = False, 0, 0
        noise = int(self._resume[0, 2].item())
        self._resume.copy_(torch.tensor([[_i32(self._origin), _i32(self._fed), noise, 0]], dtype=torch.int32))

    def getConstellationSymbols(self) -> np.ndarray:      # demodulator.cpp:827-830 (the live process() path feeds it)
        return self._constellation.copy()

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


def _i32(u: int) -> int:
    u &= 0xffffffff
    return u - (1 << 32) if u >= (1 << 31) else u


def _lib_state(name: str) -> int:
    return {"FREQ": 0, "NOISE": 1, "SNR": 2, "TIMING": 3, "PHASE": 4, "MIXER": 5, "SYMBOLS": 6}[name]
