"""Configuration types of the receive path, mirroring include/ultra/types.hpp of the
reference (same names, same numeric enum values, same defaults and helper semantics) so
that reference callers and tests read the same on this side of the boundary."""
from __future__ import annotations

from dataclasses import dataclass, replace
from enum import IntEnum


class Modulation(IntEnum):          # include/ultra/types.hpp:27-39
    DBPSK = 0
    BPSK = 1
    DQPSK = 2
    QPSK = 3
    D8PSK = 4
    QAM8 = 5
    QAM16 = 6
    QAM32 = 7
    QAM64 = 8
    QAM256 = 10


class CodeRate(IntEnum):            # include/ultra/types.hpp:91-100
    R1_4 = 0
    R1_3 = 1
    R1_2 = 2
    R2_3 = 3
    R3_4 = 4
    R5_6 = 5
    R7_8 = 6


class CyclicPrefixMode(IntEnum):    # include/ultra/types.hpp:76-80
    SHORT = 0
    MEDIUM = 1
    LONG = 2


class Entry(IntEnum):               # include/ultra_hip.h ultra_hip_entry
    SYNCED = 0
    PRESYNCED = 1


_BITS = {Modulation.DBPSK: 1, Modulation.BPSK: 1, Modulation.DQPSK: 2, Modulation.QPSK: 2,
         Modulation.D8PSK: 3, Modulation.QAM8: 3, Modulation.QAM16: 4, Modulation.QAM32: 5,
         Modulation.QAM64: 6, Modulation.QAM256: 8}
_RATE_VALUE = {CodeRate.R1_4: 0.25, CodeRate.R1_3: 0.333, CodeRate.R1_2: 0.5, CodeRate.R2_3: 0.667,
               CodeRate.R3_4: 0.75, CodeRate.R5_6: 0.833, CodeRate.R7_8: 0.875}
# getCodeParams, src/fec/ldpc_decoder.cpp:22-36 (R1_3 / R7_8 fall to the default 324/324)
_INFO_BITS = {CodeRate.R1_4: 162, CodeRate.R1_2: 324, CodeRate.R2_3: 432, CodeRate.R3_4: 486, CodeRate.R5_6: 540}
LDPC_BLOCK_SIZE = 648               # demodulator_constants.hpp:14


def getBitsPerSymbol(mod: Modulation) -> int:       # types.hpp:42-56
    return _BITS.get(Modulation(mod), 1)


def getCodeRateValue(rate: CodeRate) -> float:      # types.hpp:103-114
    return _RATE_VALUE.get(CodeRate(rate), 0.5)


def info_bits(rate: CodeRate) -> int:
    return _INFO_BITS.get(CodeRate(rate), 324)


def is_differential(mod: Modulation) -> bool:
    return Modulation(mod) in (Modulation.DBPSK, Modulation.DQPSK, Modulation.D8PSK)


@dataclass
class ModemConfig:                  # include/ultra/types.hpp:139-234 (receive-path fields)
    sample_rate: int = 48000
    center_freq: int = 1500
    fft_size: int = 512
    num_carriers: int = 30
    cp_mode: CyclicPrefixMode = CyclicPrefixMode.MEDIUM
    symbol_guard: int = 4
    pilot_spacing: int = 2
    use_pilots: bool = True
    scattered_pilots: bool = True       # never read by the reference either
    modulation: Modulation = Modulation.QPSK
    code_rate: CodeRate = CodeRate.R1_2
    adaptive_eq_enabled: bool = False   # LMS/RLS equaliser of the coherent modulations (types.hpp:170-174): off in every preset
    adaptive_eq_use_rls: bool = False
    lms_mu: float = 0.05
    rls_lambda: float = 0.99
    decision_directed: bool = True
    sync_threshold: float = 0.80        # Schmidl-Cox metric a search offset must exceed (types.hpp:188)

    def getCyclicPrefix(self) -> int:
        base = {CyclicPrefixMode.SHORT: 32, CyclicPrefixMode.MEDIUM: 48, CyclicPrefixMode.LONG: 64}[
            CyclicPrefixMode(self.cp_mode)]
        return base * (self.fft_size // 512)

    def getSymbolDuration(self) -> int:
        return self.fft_size + self.getCyclicPrefix() + self.symbol_guard

    def getSymbolRate(self) -> float:
        return self.sample_rate / self.getSymbolDuration()

    def getDataCarriers(self) -> int:
        if not self.use_pilots:
            return self.num_carriers
        pilots = (self.num_carriers + self.pilot_spacing - 1) // self.pilot_spacing
        return self.num_carriers - pilots

    def getTheoreticalThroughput(self, mod: Modulation, rate: CodeRate) -> float:
        return self.getDataCarriers() * getBitsPerSymbol(mod) * getCodeRateValue(rate) * self.getSymbolRate()

    def with_mode(self, mod: Modulation, rate: CodeRate) -> "ModemConfig":
        """What every reference harness does before constructing the modem
        (tools/test_nvis_mode.cpp:36-40): pilots iff the modulation is coherent."""
        return replace(self, modulation=Modulation(mod), code_rate=CodeRate(rate),
                       use_pilots=not is_differential(mod))


class presets:                      # include/ultra/types.hpp:262-367
    @staticmethod
    def conservative() -> ModemConfig:
        return ModemConfig(cp_mode=CyclicPrefixMode.LONG, symbol_guard=8, pilot_spacing=2,
                           modulation=Modulation.QPSK, code_rate=CodeRate.R1_2)

    @staticmethod
    def balanced() -> ModemConfig:
        return ModemConfig(cp_mode=CyclicPrefixMode.MEDIUM, symbol_guard=4, pilot_spacing=2,
                           modulation=Modulation.QAM64, code_rate=CodeRate.R3_4)

    @staticmethod
    def turbo() -> ModemConfig:
        return ModemConfig(cp_mode=CyclicPrefixMode.SHORT, symbol_guard=0, pilot_spacing=2,
                           modulation=Modulation.QAM256, code_rate=CodeRate.R5_6)

    @staticmethod
    def high_throughput() -> ModemConfig:
        return ModemConfig(fft_size=1024, num_carriers=59, cp_mode=CyclicPrefixMode.MEDIUM, symbol_guard=0,
                           pilot_spacing=4, modulation=Modulation.QAM16, code_rate=CodeRate.R2_3)

    @staticmethod
    def nvis_mode() -> ModemConfig:
        return ModemConfig(fft_size=1024, num_carriers=59, cp_mode=CyclicPrefixMode.MEDIUM, symbol_guard=0,
                           use_pilots=False, pilot_spacing=2, modulation=Modulation.DQPSK,
                           code_rate=CodeRate.R3_4)
