"""HipOfdmWaveform — the receive half of ultra::IWaveform (src/waveform/waveform_interface.hpp:
47-157) on the HIP path, shaped like OFDMChirpWaveform (src/waveform/ofdm_chirp_waveform.cpp):
an external synchroniser supplies timing + CFO, process() runs the presynced entry.

detectSync runs the dual-chirp detection kernel (scope row f4).  Transmit-side members
(generatePreamble / modulate) are outside the receive path; they raise NotImplementedError
rather than pretending."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from .ofdm import OFDMDemodulator
from .types import CodeRate, LDPC_BLOCK_SIZE, ModemConfig, Modulation, getBitsPerSymbol, is_differential


@dataclass
class SyncResult:                    # waveform_interface.hpp:33-40
    detected: bool = False
    start_sample: int = -1
    correlation: float = 0.0
    cfo_hz: float = 0.0
    snr_estimate: float = 0.0
    has_training: bool = False


class HipOfdmWaveform:
    def __init__(self, config: ModemConfig = None, device=None):
        self._config = self._chirp_config(config or ModemConfig())
        self._device = device
        self._cfo_hz = 0.0
        self._last_cfo = 0.0
        self._training_start_sample = 0
        self._soft_bits = np.zeros(0, np.float32)
        self._synced = False
        self._init_components()

    @staticmethod
    def _chirp_config(config: ModemConfig) -> ModemConfig:
        """OFDMChirpWaveform::OFDMChirpWaveform(config) / configure (ofdm_chirp_waveform.cpp:20-31,67-84): the chirp mode is
        differential and pilot-free whatever the configuration says."""
        import copy
        c = copy.copy(config)
        if c.modulation not in (Modulation.DBPSK, Modulation.DQPSK, Modulation.D8PSK):
            c.modulation = Modulation.DQPSK
        c.use_pilots = False
        return c

    def _init_components(self):
        self._demod = OFDMDemodulator(self._config, device=self._device)

    # -- identification / configuration ------------------------------------
    def getName(self) -> str:
        return "OFDM_HIP"

    def configure(self, mod: Modulation, rate: CodeRate) -> None:   # ofdm_chirp/cox_waveform.cpp configure()
        self._config = self._chirp_config(self._config.with_mode(mod, rate))
        self._init_components()

    def getModulation(self) -> Modulation:
        return self._config.modulation

    def getCodeRate(self) -> CodeRate:
        return self._config.code_rate

    def setFrequencyOffset(self, cfo_hz: float) -> None:
        self._cfo_hz = float(cfo_hz)

    def getFrequencyOffset(self) -> float:
        return self._cfo_hz

    # -- RX ------------------------------------------------------------------
    def detectSync(self, samples, result: SyncResult, threshold: float = 0.3) -> bool:
        """OFDMChirpWaveform::detectSync (ofdm_chirp_waveform.cpp:129-172) on the device."""
        from .types import Entry
        x = np.ascontiguousarray(samples, dtype=np.float32).reshape(1, -1)
        out = self._demod.context(Entry.PRESYNCED).chirp_sync(x, threshold)
        result.detected = bool(out["detected"].item())
        result.correlation = float(out["correlation"].item())
        result.cfo_hz = float(out["cfo_hz"].item())
        result.has_training = True
        if result.detected:
            self._synced = True
            self._last_cfo = result.cfo_hz
            result.start_sample = int(out["start_sample"].item())
            self._training_start_sample = result.start_sample
        return result.detected

    def accept_sync(self, result: SyncResult) -> None:
        """Take the result of an external synchroniser (what detectSync would have filled in)."""
        self._synced = bool(result.detected)
        self._cfo_hz = float(result.cfo_hz)
        self._training_start_sample = max(int(result.start_sample), 0)

    def process(self, samples) -> bool:      # OFDMChirpWaveform::process, ofdm_chirp_waveform.cpp:174-215
        sr = self._config.sample_rate
        # float initial_phase_rad = -2.0f * M_PI * cfo_hz_ * training_start_sample_ / sample_rate (double expr)
        phase = np.float32((((-2.0 * math.pi) * float(np.float32(self._cfo_hz))) * float(self._training_start_sample)) / float(sr))
        while float(phase) > math.pi:
            phase = np.float32(float(phase) - 2.0 * math.pi)
        while float(phase) < -math.pi:
            phase = np.float32(float(phase) + 2.0 * math.pi)
        self._demod.reset()
        self._demod.setFrequencyOffsetWithPhase(self._cfo_hz, float(phase))
        ready = self._demod.processPresynced(samples, 2)
        if ready:
            chunks = []
            while self._demod.hasPendingData():
                c = self._demod.getSoftBits()
                if c.size == 0:
                    break
                chunks.append(c)
            self._soft_bits = np.concatenate(chunks) if chunks else np.zeros(0, np.float32)
        return ready

    def getSoftBits(self) -> np.ndarray:
        out, self._soft_bits = self._soft_bits, np.zeros(0, np.float32)
        return out

    def reset(self) -> None:
        self._demod.reset()
        self._soft_bits = np.zeros(0, np.float32)
        self._synced = False

    def isSynced(self) -> bool:              # ofdm_chirp_waveform.cpp:232-234
        return self._synced or self._demod.isSynced()

    def hasData(self) -> bool:
        return self._soft_bits.size > 0 or self._demod.hasPendingData()

    def estimatedSNR(self) -> float:
        return self._demod.getEstimatedSNR()

    def estimatedCFO(self) -> float:         # ofdm_chirp_waveform.cpp:244-252
        if abs(self._last_cfo) > 0.1:
            return self._last_cfo
        return self._demod.getFrequencyOffset()

    # -- geometry (ofdm_chirp_waveform.cpp:266-331) ----------------------------
    def getCarrierCount(self) -> int:
        return self._config.num_carriers

    def getSamplesPerSymbol(self) -> int:
        return self._config.getSymbolDuration()

    def getPreambleSamples(self) -> int:
        """[up chirp 500 ms][gap 100 ms][down chirp][gap] (ChirpSync::getTotalSamples, chirp_sync.hpp:534-544) + two training
        symbols (ofdm_chirp_waveform.cpp:304-309) — what RxPipeline::tryProcessBuffer sizes its search by."""
        fs = np.float32(self._config.sample_rate)
        chirp, gap = int(fs * np.float32(500.0) / np.float32(1000.0)), int(fs * np.float32(100.0) / np.float32(1000.0))
        return 2 * chirp + 2 * gap + 2 * self.getSamplesPerSymbol()

    def _bits_per_carrier(self) -> int:
        return {Modulation.DBPSK: 1, Modulation.D8PSK: 3}.get(Modulation(self._config.modulation), 2)

    def getMinSamplesForFrame(self) -> int:
        bits_per_symbol = self._config.num_carriers * self._bits_per_carrier()          # every carrier is data
        data_symbols = (LDPC_BLOCK_SIZE + bits_per_symbol - 1) // bits_per_symbol
        return 2 * self.getSamplesPerSymbol() + data_symbols * self.getSamplesPerSymbol()

    def getThroughput(self, rate: CodeRate) -> float:
        c = self._config
        ratio = {CodeRate.R1_4: 0.25, CodeRate.R1_3: 0.333, CodeRate.R1_2: 0.5, CodeRate.R2_3: 0.667,
                 CodeRate.R3_4: 0.75, CodeRate.R5_6: 0.833}.get(CodeRate(rate), 0.5)
        return (c.sample_rate / self.getSamplesPerSymbol()) * c.num_carriers * self._bits_per_carrier() * ratio

    def getStatusString(self) -> str:
        c = self._config
        return f"OFDM-HIP {c.num_carriers} carriers, {Modulation(c.modulation).name} {CodeRate(c.code_rate).name}" + \
            (" (pilots)" if c.use_pilots else "")

    # -- TX: not part of the receive hot path ---------------------------------
    def generatePreamble(self):
        raise NotImplementedError("transmit side is outside the built receive path")

    def modulate(self, encoded_data):
        raise NotImplementedError("transmit side is outside the built receive path")
