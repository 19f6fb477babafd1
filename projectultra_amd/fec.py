"""LDPCDecoder — host-side mirror of ultra::LDPCDecoder (include/ultra/fec.hpp:48-77)
running on the HIP batch kernel.  Same method names, argument meaning and failure
behaviour (empty input -> empty result, lastDecodeSuccess() False) as the reference,
so tests written against the reference class read the same here."""
from __future__ import annotations

import numpy as np

from .engine import ReceiveContext
from .types import CodeRate, LDPC_BLOCK_SIZE, ModemConfig, info_bits


class LDPCDecoder:
    def __init__(self, rate: CodeRate, device=None):
        self._device = device
        self._max_iterations = 50            # Impl::max_iterations default (ldpc_decoder.cpp:43)
        self._last_success = False
        self._last_iters = 0
        self._ctx = None
        self.setRate(rate)

    # -- reference API ---------------------------------------------------
    def setRate(self, rate: CodeRate) -> None:
        self._rate = CodeRate(rate)
        self._rebuild()

    def getRate(self) -> CodeRate:
        return self._rate

    def setMaxIterations(self, max_iter: int) -> None:
        self._max_iterations = int(max_iter)
        self._rebuild()

    def lastDecodeSuccess(self) -> bool:
        return self._last_success

    def lastIterations(self) -> int:
        return self._last_iters

    def decode(self, coded_data: bytes) -> bytes:
        """Hard-bit input: +-6.0 LLRs (ldpc_decoder.cpp:267-281)."""
        bits = np.unpackbits(np.frombuffer(bytes(coded_data), np.uint8))
        return self.decodeSoft(np.where(bits == 1, -6.0, 6.0).astype(np.float32))

    def decodeSoft(self, llrs) -> bytes:
        """Bit-level multi-block semantics of LDPCDecoder::decodeSoft (ldpc_decoder.cpp:283-428)."""
        llrs = np.ascontiguousarray(llrs, dtype=np.float32).reshape(-1)
        if llrs.size == 0:
            self._last_success = False
            return b""
        n, k = LDPC_BLOCK_SIZE, info_bits(self._rate)
        nblocks = -(-llrs.size // n)
        padded = np.zeros(nblocks * n, np.float32)
        padded[:llrs.size] = llrs                       # short / tail blocks are zero padded
        r = self.decode_batch(padded.reshape(nblocks, n))
        ok, iters = r["ok"], r["iters"]
        has_tail = (llrs.size % n) != 0 and nblocks > 1
        # multi-block: success = all full blocks; a tail block decoded through decodeBP
        # overwrites last_success with its own result (ldpc_decoder.cpp:396-407)
        self._last_success = bool(ok[-1]) if (nblocks == 1 or has_tail) else bool(ok.all())
        self._last_iters = int(iters[-1])
        if nblocks == 1:
            return r["bytes"][0].tobytes()
        bits = np.unpackbits(r["bytes"], axis=1)[:, :k].reshape(-1)
        return np.packbits(bits).tobytes()

    # -- batch API (what the reference loops over) ------------------------
    def decode_batch(self, llr, want_total: bool = False, to_host: bool = True):
        """[n][648] LLRs -> dict(bytes, iters, ok[, llr_total]); numpy arrays when to_host."""
        r = self._ctx.ldpc_decode(llr, want_total=want_total)
        if to_host:
            self._ctx.synchronize()
            r = {k: v.cpu().numpy() for k, v in r.items()}
        return r

    def setDeinterleave(self, bits_per_symbol: int) -> None:
        """Fuse RxPipeline's per-codeword ChannelInterleaver(bits_per_symbol, 648)::deinterleave
        (rx_pipeline.cpp:24-31,475-491) into the decoder's LLR load; 0 switches it off."""
        self._deinterleave = int(bits_per_symbol)
        self._ctx.set_deinterleave(self._deinterleave)

    def setDeinterleaveTable(self, index) -> None:
        """Fuse any 648-entry permutation (out[j] = in[index[j]], e.g. Interleaver(6, 108).permutation) into the
        decoder's LLR load; None switches it off."""
        self._deinterleave_table = None if index is None else np.ascontiguousarray(index, np.uint16).copy()
        self._ctx.set_deinterleave_table(self._deinterleave_table)

    # ---------------------------------------------------------------------
    def _rebuild(self):
        if self._ctx is not None:
            self._ctx.close()
        cfg = ModemConfig(code_rate=self._rate)
        self._ctx = ReceiveContext(cfg, max_iterations=self._max_iterations, device=self._device)
        if getattr(self, "_deinterleave", 0):
            self._ctx.set_deinterleave(self._deinterleave)
        if getattr(self, "_deinterleave_table", None) is not None:
            self._ctx.set_deinterleave_table(self._deinterleave_table)

    @property
    def context(self) -> ReceiveContext:
        return self._ctx


class Interleaver:
    """Host mirror of ultra::Interleaver (include/ultra/fec.hpp, src/fec/ldpc_decoder.cpp:454-540): row-column transpose,
    permutation[i] = (i % cols) * rows + i // cols; interleave(soft): out[permutation[i]] = in[i]; deinterleave(soft):
    out[i] = in[permutation[i]].  The receive side runs fused on the GPU: LDPCDecoder.setDeinterleaveTable(il.permutation)."""

    def __init__(self, rows: int, cols: int):
        self.rows, self.cols = int(rows), int(cols)
        i = np.arange(self.rows * self.cols, dtype=np.int64)
        self.permutation = (i % self.cols) * self.rows + i // self.cols

    def interleave(self, soft_bits) -> np.ndarray:            # :519-527
        x = np.ascontiguousarray(soft_bits, dtype=np.float32).reshape(-1)
        out = np.zeros(x.size, np.float32)
        n = min(x.size, self.permutation.size)
        out[self.permutation[:n]] = x[:n]
        return out

    def deinterleave(self, soft_bits) -> np.ndarray:          # :530-540
        x = np.ascontiguousarray(soft_bits, dtype=np.float32).reshape(-1)
        out = np.zeros(x.size, np.float32)
        n = min(x.size, self.permutation.size)
        out[:n] = x[self.permutation[:n]]
        return out


class ChannelInterleaver:
    """Host mirror of ultra::ChannelInterleaver (include/ultra/fec.hpp:120-142,
    src/fec/ldpc_decoder.cpp:547-680): permutation[i] = (i * step) % total_bits with the coprime step of
    findCoprimeStep.  The receive side of it runs on the GPU (ReceiveContext.set_deinterleave); this
    class serves transmit-side stimulus and the tests."""

    def __init__(self, bits_per_symbol: int, total_bits: int = LDPC_BLOCK_SIZE):
        self.bits_per_symbol, self.total_bits = int(bits_per_symbol), int(total_bits)
        self.step = self._find_coprime_step(self.bits_per_symbol, self.total_bits)
        self.symbol_separation = max(1, self.step // self.bits_per_symbol)
        self.permutation = (np.arange(self.total_bits, dtype=np.int64) * self.step) % self.total_bits
        self.inverse_permutation = np.empty_like(self.permutation)
        self.inverse_permutation[self.permutation] = np.arange(self.total_bits)

    @staticmethod
    def _find_coprime_step(n: int, total: int) -> int:      # ldpc_decoder.cpp:549-573
        from math import gcd
        target = n * 3
        if target >= total:
            target = total // 2
        for step in range(target, total):
            if gcd(step, total) == 1:
                return step
        for step in range(n + 1, total):
            if gcd(step, total) == 1:
                return step
        return n + 1

    def getSymbolSeparation(self) -> int:
        return self.symbol_separation

    def interleave(self, soft_bits) -> np.ndarray:           # :598-607  output[permutation[i]] = in[i]
        x = np.ascontiguousarray(soft_bits, dtype=np.float32).reshape(-1)
        n = min(x.size, self.total_bits)
        out = np.zeros(self.total_bits, np.float32)
        out[self.permutation[:n]] = x[:n]
        return out

    def deinterleave(self, soft_bits) -> np.ndarray:         # :609-617  output[inverse_permutation[i]] = in[i]
        x = np.ascontiguousarray(soft_bits, dtype=np.float32).reshape(-1)
        n = min(x.size, self.total_bits)
        out = np.zeros(self.total_bits, np.float32)
        out[self.inverse_permutation[:n]] = x[:n]
        return out
