// hip_waveform_factory.cpp — the creation half of ultra::WaveformFactory (src/waveform/waveform_factory.hpp:18-28,
// waveform_factory.cpp:11-70) handing out the MI355X adapters.
//
// Every caller in the reference obtains its waveforms — TRANSMIT and receive — through WaveformFactory::create(mode, config)
// (src/gui/modem/modem_engine.cpp:92-94,604; tools/test_hf_modem.cpp:408,565).  This file defines the three creation
// functions with the reference's signatures:
//     OFDM_COX   -> ultra_hip::HipOfdmCoxWaveform   (Schmidl-Cox search + demodulation on the GPU; TX = the reference's OFDMModulator)
//     OFDM_CHIRP -> ultra_hip::HipOfdmWaveform      (dual-chirp detection + presynced demodulation on the GPU; TX = the
//                                                    reference's OFDMChirpWaveform: chirp generator + modulator)
//     MC_DPSK, AUTO, MFSK -> the reference's own MCDPSKWaveform (out of scope); OTFS -> the OFDM_COX adapter, as the
//     reference falls back to OFDM_COX there; an unknown mode -> nullptr
// Both adapters are complete ultra::IWaveform implementations; nothing they implement throws.
//
// Build (inside the reference tree):
//     g++ -std=c++20 -DULTRA_HIP_WITH_REFERENCE -I<ref>/include -I<ref>/src -I<repo>/include -c hip_waveform_factory.cpp
// and link it in place of src/waveform/waveform_factory.cpp when only creation is used (every caller above), or beside it with
// that file's three creation functions compiled out (the `#ifndef ULTRA_USE_HIP` guard) — its mode
// recommendation tables (recommendMode, getMinSNR, ...) are policy, not receive path, and stay the reference's.
#include <cstdlib>

#include "ultra_hip_waveform.hpp"
#include "waveform/mc_dpsk_waveform.hpp"
#include "waveform/waveform_factory.hpp"

namespace ultra {

namespace {
int hip_device() {
    const char* e = std::getenv("ULTRA_HIP_DEVICE");
    return (e && *e) ? std::atoi(e) : 0;
}
}  // namespace

WaveformPtr WaveformFactory::create(protocol::WaveformMode mode) {
    using protocol::WaveformMode;
    switch (mode) {
        case WaveformMode::OFDM_COX:
        case WaveformMode::OTFS_EQ:                                   // the reference has no OTFS waveform either: OFDM_COX (:28-32)
        case WaveformMode::OTFS_RAW:
            return std::make_unique<ultra_hip::HipOfdmCoxWaveform>(ultra_hip::HipOfdmCoxWaveform::defaultConfig(), hip_device());
        case WaveformMode::OFDM_CHIRP: {
            ModemConfig c;                                            // OFDMChirpWaveform::OFDMChirpWaveform() (ofdm_chirp_waveform.cpp:10-18)
            c.fft_size = 512; c.num_carriers = 30; c.modulation = Modulation::DQPSK; c.code_rate = CodeRate::R1_2; c.use_pilots = false;
            return std::make_unique<ultra_hip::HipOfdmWaveform>(c, hip_device());
        }
        case WaveformMode::MC_DPSK:
        case WaveformMode::AUTO:
        case WaveformMode::MFSK:
            return std::make_unique<MCDPSKWaveform>();
        default:
            return nullptr;
    }
}

WaveformPtr WaveformFactory::create(protocol::WaveformMode mode, const ModemConfig& config) {
    using protocol::WaveformMode;
    switch (mode) {
        case WaveformMode::OFDM_COX:
            return std::make_unique<ultra_hip::HipOfdmCoxWaveform>(config, hip_device());
        case WaveformMode::OFDM_CHIRP:
            return std::make_unique<ultra_hip::HipOfdmWaveform>(config, hip_device());
        case WaveformMode::MC_DPSK: {
            MultiCarrierDPSKConfig mc;                                // only the sample rate travels (:47-52)
            mc.sample_rate = static_cast<float>(config.sample_rate);
            return std::make_unique<MCDPSKWaveform>(mc);
        }
        default:
            return create(mode);
    }
}

WaveformPtr WaveformFactory::createMCDPSK(int num_carriers) { return std::make_unique<MCDPSKWaveform>(num_carriers); }

}  // namespace ultra
