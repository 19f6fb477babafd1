// tools/ldpc_tplan_check.cpp — host check of the totals-kernel plan (csrc/host_tables.h build_ldpc_tplan on the embedded
// placement csrc/ldpc_placement.h): the plan is valid for all six codes (round 4: ldpc_placement_low.h for R1/4, R1/3, R1/2), and a lane-by-lane CPU emulation of
// ldpc_totals_kernel.h over that plan (T / R arrays, gather addresses, pad words, the parity verdict at the top of the next
// iteration) gives exactly the reference's decodeBP result (src/fec/ldpc_decoder.cpp:153-259) — iterations, success, bits.
//   g++ -O2 -std=c++17 -ffp-contract=off -Iprojectultra_amd/csrc tools/ldpc_tplan_check.cpp -o /tmp/tpc && /tmp/tpc
#include "host_tables.h"
#include <cstdio>
#include <cstring>
#include <random>
#include <cfloat>
using namespace ultra_hip;
// reference decodeBP (ldpc_decoder.cpp:153-259), direct
static void ref_decode(const LdpcConst& L, int maxit, const float* llr, std::vector<uint8_t>& hard, int& iters, int& ok) {
    int n = L.n, m = L.m; std::vector<float> v2c(L.edges), c2v(L.edges, 0.f), tot(n);
    for (int i = 0; i < m; ++i) for (int e = L.row_ptr[i]; e < L.row_ptr[i+1]; ++e) v2c[e] = llr[L.col[e]];
    for (int j = 0; j < n; ++j) tot[j] = llr[j];
    ok = 0; int it;
    for (it = 0; it < maxit; ++it) {
        for (int i = 0; i < m; ++i) for (int e = L.row_ptr[i]; e < L.row_ptr[i+1]; ++e) {
            float sign = 1.0f, mn = FLT_MAX;
            for (int e2 = L.row_ptr[i]; e2 < L.row_ptr[i+1]; ++e2) if (e2 != e) { float msg = v2c[e2]; if (msg < 0) sign = -sign; float a = std::fabs(msg); if (a < mn) mn = a; }
            c2v[e] = sign * mn * 0.75f; }
        for (int j = 0; j < n; ++j) tot[j] = llr[j];
        for (int i = 0; i < m; ++i) for (int e = L.row_ptr[i]; e < L.row_ptr[i+1]; ++e) tot[L.col[e]] += c2v[e];
        for (int i = 0; i < m; ++i) for (int e = L.row_ptr[i]; e < L.row_ptr[i+1]; ++e) { float x = tot[L.col[e]] - c2v[e]; v2c[e] = std::max(-50.0f, std::min(50.0f, x)); }
        bool good = true;
        for (int i = 0; i < m && good; ++i) { int s = 0; for (int e = L.row_ptr[i]; e < L.row_ptr[i+1]; ++e) s ^= (tot[L.col[e]] < 0); if (s) good = false; }
        if (good) { ok = 1; break; }
    }
    iters = it; hard.resize(n); for (int j = 0; j < n; ++j) hard[j] = tot[j] < 0;
}
// totals algorithm exactly as the kernel, lanes simulated
static void sim_decode(const LdpcConst& L, const LdpcTPlan& P, const float* llr, std::vector<uint8_t>& hard, int& iters, int& ok) {
    std::vector<unsigned char> lds(P.lds_bytes + 64, 0);
    auto F = [&](unsigned off) -> float& { return *reinterpret_cast<float*>(&lds[off]); };
    for (int b = 0; b < 32; ++b) { F(P.t_pad + 4 * b) = FLT_MAX; F(P.r_pad + 4 * b) = -0.0f; }      // one pad word per bank
    const int RR = P.row_rounds, VR = P.var_rounds;
    auto Sr = [&](int r) { return (int)((P.row_prof >> (4 * r)) & 15ull); };
    auto Dr = [&](int r) { return (int)((P.var_prof >> (4 * r)) & 15ull); };
    std::vector<float> c2v(RR * 64 * 7, 0.f);
    for (int s = 0; s < VR * 64; ++s) if (P.var_id[s] != 0xFFFF) F(s * 4) = llr[P.var_id[s]];
    int it = 0; ok = 0;
    for (;;) {
        const float cap = it == 0 ? FLT_MAX : 50.0f;
        bool bad = false;
        std::vector<std::pair<unsigned, float>> stores;
        for (int s = 0; s < RR * 64; ++s) { if (P.row_check[s] == 0xFFFF) continue;
            const int S = Sr(s / 64);                                   // information-edge slots of the round; the parity edge is edge S
            float* c = &c2v[s * 7]; float v[7]; float lp = llr[L.k + P.row_check[s]];
            float tp = lp + c[S]; bool synd = tp < 0;
            for (int t = 0; t < S; ++t) { float tot = F(P.row_taddr[s * 6 + t]); synd ^= (tot < 0); v[t] = tot - c[t]; }
            v[S] = tp - c[S]; bad |= synd;
            bool ng[7], par = false; for (int t = 0; t <= S; ++t) { ng[t] = v[t] < 0; par ^= ng[t]; }
            for (int t = 0; t <= S; ++t) { float mn = cap; for (int u = 0; u <= S; ++u) if (u != t) { float a = std::fabs(v[u]); if (a < mn) mn = a; } float mag = mn * 0.75f; c[t] = (par != ng[t]) ? -mag : mag; }
            for (int t = 0; t < S; ++t) stores.push_back({(unsigned)(P.r_base + ((P.plane_base[s / 64] + t) * 64 + s % 64) * 4), c[t]});
        }
        for (auto& st : stores) F(st.first) = st.second;
        if (it > 0 && !bad) { ok = 1; --it; break; }
        if (it >= P.max_iterations) break;
        std::vector<std::pair<unsigned, float>> ts;
        for (int s = 0; s < VR * 64; ++s) { if (P.var_id[s] == 0xFFFF) continue; float tot = llr[P.var_id[s]]; for (int q = 0; q < Dr(s / 64); ++q) tot += F(P.var_caddr[s * kTPlanDmax + q]); ts.push_back({(unsigned)(s * 4), tot}); }
        for (auto& st : ts) F(st.first) = st.second;
        ++it;
    }
    iters = ok ? it : P.max_iterations;
    hard.resize(L.n); for (int j = 0; j < L.k; ++j) { unsigned sl = P.var_slot_of[j]; hard[j] = ((sl != 0xFFFF) ? F(sl * 4) : llr[j]) < 0; }
}
int main() {
    std::mt19937 rng(3); std::normal_distribution<float> N(0, 1);
    int total_bad = 0;
    for (uint32_t rate = 0; rate < 6; ++rate) {
        LdpcConst L; build_ldpc(rate, 50, L); LdpcTPlan P; const int rc = build_ldpc_tplan(L, rate, P);
        if (rc != 0 || !P.valid) { printf("rate %u: plan INVALID rc %d\n", rate, rc); ++total_bad; continue; }
        printf("rate %u: plan valid, %d row rounds (profile 0x%llx, %d R planes), %d variable rounds (profile 0x%llx), LDS %d B, residual gather collisions cost %d cycles per iteration\n",
               rate, P.row_rounds, (unsigned long long)P.row_prof, P.n_planes, P.var_rounds, (unsigned long long)P.var_prof, P.lds_bytes, P.extra_cycles);
        int bad = 0, n = 200; double si = 0, ri = 0;
        for (int c = 0; c < n; ++c) {
            const float base_sig[6] = {2.0f, 1.3f, 1.0f, 0.75f, 0.6f, 0.5f};
            std::vector<float> llr(L.n); float sig = base_sig[rate] * (0.8f + 0.4f * (c % 3) / 2.0f);
            for (auto& x : llr) x = 2.0f * (1.0f + sig * N(rng)) / (sig * sig);
            std::vector<uint8_t> h1, h2; int i1, o1, i2, o2;
            ref_decode(L, 50, llr.data(), h1, i1, o1); sim_decode(L, P, llr.data(), h2, i2, o2);
            bool same = i1 == i2 && o1 == o2; for (int j = 0; j < L.k && same; ++j) same = h1[j] == h2[j];
            if (!same) { if (bad < 5) printf("rate %u cw %d: ref iters %d ok %d, sim iters %d ok %d\n", rate, c, i1, o1, i2, o2); ++bad; }
            si += i2; ri += i1;
        }
        printf("rate %u: emulation vs reference decodeBP: %d mismatches of %d (mean iterations %.2f / %.2f)\n", rate, bad, n, ri / n, si / n);
        total_bad += bad;
    }
    return total_bad ? 1 : 0;
}
