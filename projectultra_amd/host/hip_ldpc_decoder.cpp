// hip_ldpc_decoder.cpp — link-time replacement of ultra::LDPCDecoder by the MI355X path.
//
// ultra::LDPCDecoder is a pimpl class (include/ultra/fec.hpp:48-77) that every caller constructs directly
// (tools/test_nvis_mode.cpp:46, tools/test_mode_snr.cpp:36, src/gui/modem/rx_pipeline.cpp:348-444, src/modem/modem.cpp:79-112).
// This translation unit DEFINES it: Impl is ultra_hip::HipLDPCDecoder (include/ultra_hip_waveform.hpp over
// ultra_hip_ldpc_decode_batch — the scaled min-sum decoder of src/fec/ldpc_decoder.cpp:153-428 as a HIP kernel).  Compile it
// inside the reference tree
//
//     g++ -std=c++20 -DULTRA_HIP_WITH_REFERENCE -I<ref>/include -I<ref>/src -I<repo>/include -c hip_ldpc_decoder.cpp
//
// and link it INSTEAD OF src/fec/ldpc_decoder.cpp, with -lultra_hip.
//
// src/fec/ldpc_decoder.cpp also holds the two interleavers of include/ultra/fec.hpp:85-147 (:454-680) — plain index
// permutations on the host, used on the transmit side and by RxPipeline::deinterleaveCodewords.  They are defined here too, so
// the replaced file leaves no undefined symbol; the permutations come from the same table builders the device path fuses into
// the decoder's LLR load (ultra_hip_channel_interleaver_step, ultra_hip_set_deinterleave).
//
// Device: ULTRA_HIP_DEVICE in the environment (default 0).
#include <cstdlib>
#include <numeric>

#include "ultra/fec.hpp"
#include "ultra_hip_waveform.hpp"

namespace ultra {

namespace {
int hip_device() {
    const char* e = std::getenv("ULTRA_HIP_DEVICE");
    return (e && *e) ? std::atoi(e) : 0;
}

// MSB-first bit view of a byte string, cut or zero-padded to n bits
std::vector<uint8_t> unpack_bits(ByteSpan data, size_t n) {
    std::vector<uint8_t> bits(n, 0);
    const size_t have = std::min(n, data.size() * 8);
    for (size_t i = 0; i < have; ++i) bits[i] = (data[i >> 3] >> (7 - (i & 7))) & 1u;
    return bits;
}
Bytes pack_bits(const std::vector<uint8_t>& bits) {
    Bytes out((bits.size() + 7) / 8, 0);
    for (size_t i = 0; i < bits.size(); ++i)
        if (bits[i]) out[i >> 3] |= uint8_t(0x80u >> (i & 7));
    return out;
}
// scatter: out[index[i]] = in[i]; gather: out[i] = in[index[i]] — over the first n entries, out-of-range targets skipped
template <class T>
std::vector<T> scatter(const std::vector<size_t>& index, std::span<const T> in, size_t out_len, size_t n) {
    std::vector<T> out(out_len, T{});
    for (size_t i = 0; i < n; ++i)
        if (index[i] < out_len) out[index[i]] = in[i];
    return out;
}
template <class T>
std::vector<T> gather(const std::vector<size_t>& index, std::span<const T> in, size_t out_len, size_t n) {
    std::vector<T> out(out_len, T{});
    for (size_t i = 0; i < n; ++i)
        if (index[i] < in.size()) out[i] = in[index[i]];
    return out;
}
}  // namespace

// ---------------------------------------------------------------------------------------------
struct LDPCDecoder::Impl {
    ultra_hip::HipLDPCDecoder d;
    explicit Impl(CodeRate rate) : d(rate, hip_device()) {}
};

LDPCDecoder::LDPCDecoder(CodeRate rate) : impl_(std::make_unique<Impl>(rate)) {}
LDPCDecoder::~LDPCDecoder() = default;

// No exception leaves the class: a failing C-ABI call is reported on stderr and the decode counts as failed (empty result,
// lastDecodeSuccess() false) — the reference's decoder does not throw, and RxPipeline calls it on the audio thread.
Bytes LDPCDecoder::decode(ByteSpan coded_data) {
    return ultra_hip::detail::guarded<Bytes>("LDPCDecoder::decode", Bytes{}, [&] {
        return impl_->d.decode(std::span<const uint8_t>(coded_data.data(), coded_data.size())); });
}
Bytes LDPCDecoder::decodeSoft(std::span<const float> llrs) {
    return ultra_hip::detail::guarded<Bytes>("LDPCDecoder::decodeSoft", Bytes{}, [&] { return impl_->d.decodeSoft(llrs); });
}
bool LDPCDecoder::lastDecodeSuccess() const { return impl_->d.lastDecodeSuccess(); }
int LDPCDecoder::lastIterations() const { return impl_->d.lastIterations(); }
void LDPCDecoder::setRate(CodeRate rate) { impl_->d.setRate(rate); }
CodeRate LDPCDecoder::getRate() const { return impl_->d.getRate(); }
void LDPCDecoder::setMaxIterations(int max_iter) { impl_->d.setMaxIterations(max_iter); }

// ---------------------------------------------------------------------------------------------
// Interleaver(rows, cols): bit i = (row, col) of a row-major rows x cols block goes to position col * rows + row.
Interleaver::Interleaver(size_t rows, size_t cols) : rows_(rows), cols_(cols), permutation_(rows * cols) {
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c) permutation_[r * cols + c] = c * rows + r;
}

Bytes Interleaver::interleave(ByteSpan data) {
    const size_t n = permutation_.size();
    const std::vector<uint8_t> bits = unpack_bits(data, n);
    return pack_bits(scatter<uint8_t>(permutation_, bits, n, n));
}
Bytes Interleaver::deinterleave(ByteSpan data) {
    const size_t n = permutation_.size();
    const std::vector<uint8_t> bits = unpack_bits(data, n);
    return pack_bits(gather<uint8_t>(permutation_, bits, n, n));
}
std::vector<float> Interleaver::interleave(std::span<const float> soft_bits) {
    const size_t n = soft_bits.size();
    return scatter<float>(permutation_, soft_bits, n, std::min(n, permutation_.size()));
}
std::vector<float> Interleaver::deinterleave(std::span<const float> soft_bits) {
    const size_t n = soft_bits.size();
    return gather<float>(permutation_, soft_bits, n, std::min(n, permutation_.size()));
}

// ---------------------------------------------------------------------------------------------
// ChannelInterleaver(bits_per_symbol, total): position i -> (i * step) mod total with a step coprime to total that lands
// consecutive bits about three OFDM symbols apart; the step is the one the decoder kernel's fused deinterleave uses.
ChannelInterleaver::ChannelInterleaver(size_t bits_per_symbol, size_t total_bits)
    : bits_per_symbol_(bits_per_symbol), total_bits_(total_bits) {
    num_symbols_ = (total_bits + bits_per_symbol - 1) / bits_per_symbol;
    uint32_t step = 0;
    if (ultra_hip_channel_interleaver_step(static_cast<uint32_t>(bits_per_symbol), static_cast<uint32_t>(total_bits), &step) != ULTRA_HIP_OK)
        step = static_cast<uint32_t>(bits_per_symbol + 1);
    symbol_separation_ = std::max<size_t>(step / bits_per_symbol, 1);
    permutation_.resize(total_bits);
    inverse_permutation_.resize(total_bits);
    size_t at = 0;                                                   // (i * step) mod total, walked incrementally
    for (size_t i = 0; i < total_bits; ++i) {
        permutation_[i] = at;
        inverse_permutation_[at] = i;
        at += step;
        if (at >= total_bits) at %= total_bits;
    }
}

std::vector<float> ChannelInterleaver::interleave(std::span<const float> soft_bits) {
    return scatter<float>(permutation_, soft_bits, total_bits_, std::min(soft_bits.size(), total_bits_));
}
std::vector<float> ChannelInterleaver::deinterleave(std::span<const float> soft_bits) {
    return scatter<float>(inverse_permutation_, soft_bits, total_bits_, std::min(soft_bits.size(), total_bits_));
}
Bytes ChannelInterleaver::interleave(ByteSpan data) {
    const std::vector<uint8_t> bits = unpack_bits(data, total_bits_);
    return pack_bits(scatter<uint8_t>(permutation_, bits, total_bits_, total_bits_));
}
Bytes ChannelInterleaver::deinterleave(ByteSpan data) {
    const std::vector<uint8_t> bits = unpack_bits(data, total_bits_);
    return pack_bits(scatter<uint8_t>(inverse_permutation_, bits, total_bits_, total_bits_));
}

}  // namespace ultra
