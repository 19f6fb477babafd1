// host_tables.h — host-side construction of the per-configuration constant
// tables, done once per context exactly as the reference's constructors do it
// (same RNG walks, same float/double promotions, same libm calls on the host).
//
//   NCO sequence          src/dsp/filters.cpp:228-238   (NCO::NCO, NCO::next)
//   FFT twiddles          src/dsp/fft.cpp:75-82
//   carrier layout        src/ofdm/demodulator.cpp:46-69 (setupCarriers)
//   Zadoff-Chu + pilots   src/ofdm/demodulator.cpp:71-85 (generateSequences)
//   interpolation table   src/ofdm/demodulator.cpp:137-193 (buildInterpTable)
//   Tanner graph          src/fec/ldpc_decoder.cpp:64-137 (buildMatrix)
#ifndef ULTRA_HOST_TABLES_H
#define ULTRA_HOST_TABLES_H

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <random>
#include <vector>

#include "../../include/ultra_hip.h"
#include "device_types.h"
#ifndef ULTRA_LDPC_NO_PLACEMENT
#include "ldpc_placement.h"
#include "ldpc_placement_low.h"
#else       // tools/ldpc_place.cpp, which GENERATES that header, compiles without it
namespace ultra_hip {
inline bool ldpc_placement(uint32_t, const uint16_t**, int*, const uint16_t**, int*) { return false; }
inline bool ldpc_placement_low(uint32_t, const uint16_t**, int*, const uint16_t**, int*, unsigned long long*, unsigned long long*) { return false; }
}
#endif

namespace ultra_hip {

inline uint32_t bits_per_symbol(uint32_t mod) {  // include/ultra/types.hpp:42-56
    switch (mod) {
        case ULTRA_MOD_DBPSK: case ULTRA_MOD_BPSK: return 1;
        case ULTRA_MOD_DQPSK: case ULTRA_MOD_QPSK: return 2;
        case ULTRA_MOD_D8PSK: case ULTRA_MOD_QAM8: return 3;
        case ULTRA_MOD_QAM16: return 4;
        case ULTRA_MOD_QAM32: return 5;
        case ULTRA_MOD_QAM64: return 6;
        case ULTRA_MOD_QAM256: return 8;
        default: return 1;
    }
}

inline uint32_t cyclic_prefix(const ultra_hip_config& c) {  // include/ultra/types.hpp:197-208
    uint32_t base = 48;
    if (c.cp_mode == ULTRA_CP_SHORT) base = 32;
    else if (c.cp_mode == ULTRA_CP_LONG) base = 64;
    return base * (c.fft_size / 512);
}

inline void code_params(uint32_t rate, int& k, int& m) {  // src/fec/ldpc_decoder.cpp:22-36
    switch (rate) {
        case ULTRA_RATE_R1_4: k = 162; m = 486; break;
        case ULTRA_RATE_R2_3: k = 432; m = 216; break;
        case ULTRA_RATE_R3_4: k = 486; m = 162; break;
        case ULTRA_RATE_R5_6: k = 540; m = 108; break;
        case ULTRA_RATE_R1_2:
        default: k = 324; m = 324; break;
    }
}

inline float ce_margin(uint32_t mod) {  // src/ofdm/soft_demap.hpp:243-264
    switch (mod) {
        case ULTRA_MOD_D8PSK: case ULTRA_MOD_QAM8: return 1.1f;
        case ULTRA_MOD_QAM16: return 1.2f;
        case ULTRA_MOD_QAM32: return 1.5f;
        case ULTRA_MOD_QAM64: return 1.8f;
        case ULTRA_MOD_QAM256: return 2.5f;
        default: return 1.0f;
    }
}

inline bool known_modulation(uint32_t mod) {
    switch (mod) {
        case ULTRA_MOD_DBPSK: case ULTRA_MOD_BPSK: case ULTRA_MOD_DQPSK: case ULTRA_MOD_QPSK:
        case ULTRA_MOD_D8PSK: case ULTRA_MOD_QAM8: case ULTRA_MOD_QAM16: case ULTRA_MOD_QAM32:
        case ULTRA_MOD_QAM64: case ULTRA_MOD_QAM256: return true;
        default: return false;
    }
}

// Validate the configuration against what the path was built for.
inline int validate_config(const ultra_hip_config& c) {
    if (c.fft_size != 512 && c.fft_size != 1024) return ULTRA_HIP_ERR_UNSUPPORTED;
    if (c.num_carriers == 0 || c.num_carriers > (uint32_t)kMaxCarriers) return ULTRA_HIP_ERR_UNSUPPORTED;
    if (c.num_carriers + 1 >= c.fft_size) return ULTRA_HIP_ERR_INVALID_ARG;
    if (c.pilot_spacing == 0 || c.sample_rate == 0) return ULTRA_HIP_ERR_INVALID_ARG;
    // every carrier a pilot: no data carriers, nothing to demodulate (and track_pilot_kernel holds <= 32 pilots)
    if (c.use_pilots && c.pilot_spacing == 1) return ULTRA_HIP_ERR_UNSUPPORTED;
    if (c.cp_mode > ULTRA_CP_LONG || c.use_pilots > 1) return ULTRA_HIP_ERR_INVALID_ARG;
    if (!known_modulation(c.modulation)) return ULTRA_HIP_ERR_INVALID_ARG;
    // QAM8 has no mapper/demapper of its own in the reference (falls through to QPSK with a
    // 3-bit carrier budget: modulator.cpp:104-107, demodulator.cpp:352-356) — not a usable mode
    if (c.modulation == ULTRA_MOD_QAM8) return ULTRA_HIP_ERR_UNSUPPORTED;
    if (c.fft_size + 64 * (c.fft_size / 512) + c.symbol_guard > 1280) return ULTRA_HIP_ERR_UNSUPPORTED;
    if (c.code_rate > ULTRA_RATE_R5_6) return ULTRA_HIP_ERR_UNSUPPORTED;  // R7/8 has no code in the reference either
    if (c.entry > ULTRA_ENTRY_PRESYNCED) return ULTRA_HIP_ERR_INVALID_ARG;
    // MAX_SYMBOLS_BEFORE_TIMEOUT = 250: process() gives up after the 251st symbol of a frame (demodulator.cpp:683-691)
    if (c.n_data_symbols == 0 || c.n_data_symbols > 251) return ULTRA_HIP_ERR_INVALID_ARG;
    if (c.entry == ULTRA_ENTRY_PRESYNCED && c.training_symbols > 8) return ULTRA_HIP_ERR_INVALID_ARG;
    if (c.max_iterations > 1000) return ULTRA_HIP_ERR_INVALID_ARG;
    if (c.adaptive_eq_enabled > 1 || c.adaptive_eq_use_rls > 1 || c.decision_directed > 1) return ULTRA_HIP_ERR_INVALID_ARG;
    if (!(c.sync_threshold >= 0.0f)) return ULTRA_HIP_ERR_INVALID_ARG;   // 0 = the default 0.80; NaN and negatives are the caller's error
    return ULTRA_HIP_OK;
}

inline int fill_geometry(const ultra_hip_config& c, ultra_hip_geometry& g) {
    int rc = validate_config(c);
    if (rc != ULTRA_HIP_OK) return rc;
    int k, m;
    code_params(c.code_rate, k, m);
    g.cp_len = cyclic_prefix(c);
    g.symbol_samples = c.fft_size + g.cp_len + c.symbol_guard;
    uint32_t tr = (c.entry == ULTRA_ENTRY_PRESYNCED) ? c.training_symbols : 0;
    g.frame_samples = (tr + c.n_data_symbols) * g.symbol_samples;
    uint32_t n_pilot = c.use_pilots ? (c.num_carriers + c.pilot_spacing - 1) / c.pilot_spacing : 0;
    g.n_pilot_carriers = n_pilot;
    g.n_data_carriers = c.num_carriers - n_pilot;
    g.bits_per_carrier = bits_per_symbol(c.modulation);
    g.llrs_per_symbol = g.n_data_carriers * g.bits_per_carrier;
    g.llrs_per_frame = g.llrs_per_symbol * c.n_data_symbols;
    g.ldpc_n = kLdpcN;
    g.ldpc_k = (uint32_t)k;
    g.ldpc_m = (uint32_t)m;
    g.ldpc_edges = 0;  // filled by build_ldpc
    g.decoded_bytes = (uint32_t)((k + 7) / 8);
    return ULTRA_HIP_OK;
}

// ---- Tanner graph ---------------------------------------------------------
inline void build_ldpc(uint32_t rate, uint32_t max_iterations, LdpcConst& L) {
    int k, m;
    code_params(rate, k, m);
    std::mt19937 rng(0x12345678 + static_cast<int>(rate));
    std::vector<std::vector<int>> rows(m);
    const int target_check_degree = 4;
    int target_var_degree = std::max(3, (target_check_degree * m) / k);
    target_var_degree = std::min(target_var_degree, m / 2);
    const int max_check_degree = target_check_degree + 2;
    std::vector<int> check_degrees(m, 0), pool;
    for (int j = 0; j < k; ++j) {
        pool.clear();
        for (int i = 0; i < m; ++i)
            if (check_degrees[i] < max_check_degree) pool.push_back(i);
        for (size_t i = pool.size(); i > 1; --i) {       // Fisher-Yates, rng() % i from the top
            size_t pick = rng() % i;
            std::swap(pool[i - 1], pool[pick]);
        }
        int links = std::min(target_var_degree, (int)pool.size());
        for (int d = 0; d < links; ++d) { rows[pool[d]].push_back(j); check_degrees[pool[d]]++; }
    }
    for (int i = 0; i < m; ++i)
        if (rows[i].empty()) rows[i].push_back((int)(rng() % k));
    for (int i = 0; i < m; ++i) rows[i].push_back(k + i);   // identity part

    L = LdpcConst{};
    L.k = k; L.m = m; L.n = k + m; L.max_iterations = (int)max_iterations;
    L.decoded_bytes = (k + 7) / 8;
    int e = 0;
    std::vector<std::vector<int>> var_edges(k + m);
    for (int i = 0; i < m; ++i) {
        L.row_ptr[i] = (uint16_t)e;
        for (int v : rows[i]) { L.col[e] = (uint16_t)v; var_edges[v].push_back(e); ++e; }
    }
    L.row_ptr[m] = (uint16_t)e;
    L.edges = e;
    int q = 0;
    for (int j = 0; j < k + m; ++j) {
        L.var_ptr[j] = (uint16_t)q;
        for (int ed : var_edges[j]) L.var_edge[q++] = (uint16_t)ed;  // ascending check order
    }
    L.var_ptr[k + m] = (uint16_t)q;
}

// Proper edge colouring of a bipartite multigraph with maximum degree <= ncolors (Koenig):
// colour[e] in [0, ncolors), no two edges sharing an endpoint share a colour.
inline void bipartite_edge_colouring(int n_left, int n_right, const std::vector<std::pair<int, int>>& edges,
                                     int ncolors, std::vector<int>& colour) {
    std::vector<std::vector<int>> at_l(n_left, std::vector<int>(ncolors, -1)), at_r(n_right, std::vector<int>(ncolors, -1));
    colour.assign(edges.size(), -1);
    for (size_t e = 0; e < edges.size(); ++e) {
        const int u = edges[e].first, v = edges[e].second;
        int a = 0, b = 0;
        while (at_l[u][a] != -1) ++a;
        while (at_r[v][b] != -1) ++b;
        if (a != b) {
            // colour a is taken at v: flip the a/b alternating path that starts at v
            std::vector<int> path;
            int node = v, want = a;
            bool right = true;
            for (;;) {
                const int f = right ? at_r[node][want] : at_l[node][want];
                if (f == -1) break;
                path.push_back(f);
                node = right ? edges[f].first : edges[f].second;
                right = !right;
                want = (want == a) ? b : a;
            }
            for (int f : path) { at_l[edges[f].first][colour[f]] = -1; at_r[edges[f].second][colour[f]] = -1; }
            for (int f : path) {
                colour[f] = (colour[f] == a) ? b : a;
                at_l[edges[f].first][colour[f]] = f; at_r[edges[f].second][colour[f]] = f;
            }
        }
        colour[e] = a;
        at_l[u][a] = (int)e; at_r[v][a] = (int)e;
    }
}

// Execution plan for the kernel, derived from the CSR graph.
// ChannelInterleaver (src/fec/ldpc_decoder.cpp:547-596): permutation[i] = (i * step) % total with the
// step chosen by findCoprimeStep(bits_per_symbol, total) (:549-573); deinterleave writes
// out[inverse_permutation[i]] = in[i], i.e. out[j] = in[(j * step) % total].
inline uint32_t channel_interleaver_step(uint32_t bits_per_symbol, uint32_t total) {
    auto gcd = [](size_t a, size_t b) { while (b != 0) { const size_t t = b; b = a % b; a = t; } return a; };
    const size_t n = bits_per_symbol, tot = total;
    size_t target = n * 3;
    if (target >= tot) target = tot / 2;
    for (size_t step = target; step < tot; ++step) if (gcd(step, tot) == 1) return (uint32_t)step;
    for (size_t step = n + 1; step < tot; ++step) if (gcd(step, tot) == 1) return (uint32_t)step;
    return (uint32_t)(n + 1);
}

// The lane-linear layout (LdpcPlan::linear).  Every edge of the variable in lane l has LDS bank l mod 32, so the
// check step's gather of slot t over a half-wave of rows is conflict-free iff those rows' t-th edges go to
// variables of 32 different banks.  Rows are dealt into the 2 * row_rounds half-waves evenly (27 of 32 lanes for
// the codes at hand, which is the slack that makes this solvable), variables into banks under the capacity of two
// lanes per round and bank, and a short annealing run swaps variables between banks and rows between half-waves
// until no (half-wave, bank) cell holds more than six edges; a bipartite multigraph of maximum degree six is
// six-edge-colourable (Koenig), which then gives every edge of a half-wave its slot.  Deterministic (fixed seeds).
// Returns ULTRA_HIP_ERR_UNSUPPORTED when the code does not qualify or no assignment was found.
inline int build_ldpc_plan_linear(const LdpcConst& L, LdpcPlan& P) {
    P = LdpcPlan{};
    P.k = L.k; P.m = L.m; P.n = L.n; P.edges = L.edges; P.max_iterations = L.max_iterations;
    P.decoded_bytes = L.decoded_bytes;
    // rows: six information edges + parity, but for a few shorter ones, which stay in the last half-wave
    auto rdeg = [&](int i) { return (int)(L.row_ptr[i + 1] - L.row_ptr[i]); };
    int n_short = 0;
    for (int i = 0; i < L.m; ++i) {
        if (rdeg(i) < 2 || rdeg(i) > 7) return ULTRA_HIP_ERR_UNSUPPORTED;
        if (L.col[L.row_ptr[i + 1] - 1] != L.k + i) return ULTRA_HIP_ERR_UNSUPPORTED;
        if (rdeg(i) != 7) ++n_short;
    }
    if (n_short > 4) return ULTRA_HIP_ERR_UNSUPPORTED;
    for (int j = L.k; j < L.n; ++j)
        if (L.var_ptr[j + 1] - L.var_ptr[j] != 1) return ULTRA_HIP_ERR_UNSUPPORTED;
    auto vdeg = [&](int j) { return (int)(L.var_ptr[j + 1] - L.var_ptr[j]); };
    std::vector<int> act_list;
    int dmax = 0;
    for (int j = 0; j < L.k; ++j) if (vdeg(j) > 0) { act_list.push_back(j); dmax = std::max(dmax, vdeg(j)); }
    std::stable_sort(act_list.begin(), act_list.end(), [&](int a, int b) { return vdeg(a) > vdeg(b); });
    const int na = (int)act_list.size();
    const int VR = (na + 63) / 64, RR = (L.m + 63) / 64, NG = 2 * RR;
    if (dmax > 4 || VR > 9 || RR > 8 || VR * 64 > kLdpcPlanMaxActive) return ULTRA_HIP_ERR_UNSUPPORTED;
    const int per_group = (L.m + NG - 1) / NG;
    if (per_group > 32) return ULTRA_HIP_ERR_UNSUPPORTED;
    const int n_fullv = (VR - 1) * 64;                        // variables of the full rounds (sorted order)
    for (int a = 0; a < n_fullv; ++a) if (vdeg(act_list[a]) != dmax) return ULTRA_HIP_ERR_UNSUPPORTED;

    // rows of each variable (information edges only)
    std::vector<std::vector<int>> var_rows(L.k);
    for (int i = 0; i < L.m; ++i)
        for (int e = L.row_ptr[i]; e + 1 < L.row_ptr[i + 1]; ++e) var_rows[L.col[e]].push_back(i);

    std::mt19937 rng(0x1DEA5u);
    std::vector<int> bank(L.k, -1), grp(L.m);
    // initial deal: full-round variables round-robin over the banks (capacity 2 * (VR - 1) each), the last round's
    // variables over distinct banks; rows round-robin over the half-waves
    for (int a = 0; a < na; ++a) bank[act_list[a]] = (a < n_fullv) ? a % 32 : (a - n_fullv) % 32;
    {
        int dealt = 0;
        for (int i = 0; i < L.m; ++i) {
            if (rdeg(i) != 7) { grp[i] = NG - 1; continue; }
            grp[i] = dealt % NG;
            if (grp[i] == NG - 1 && dealt / NG >= per_group - n_short) grp[i] = (dealt + 1) % (NG - 1);   // room for the short rows
            ++dealt;
        }
        std::vector<int> fill(NG, 0);
        for (int i = 0; i < L.m; ++i) if (++fill[grp[i]] > 32) return ULTRA_HIP_ERR_UNSUPPORTED;
    }
    std::vector<int> cnt(NG * 32, 0);
    for (int i = 0; i < L.m; ++i)
        for (int e = L.row_ptr[i]; e + 1 < L.row_ptr[i + 1]; ++e) cnt[grp[i] * 32 + bank[L.col[e]]]++;
    auto over = [](int c) { return c > 6 ? c - 6 : 0; };
    long cost = 0;
    for (int c : cnt) cost += over(c);
    double T = 1.0;
    auto bump = [&](int g, int b, int s) { const int o = over(cnt[g * 32 + b]); cnt[g * 32 + b] += s; return over(cnt[g * 32 + b]) - o; };
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (long it = 0; cost > 0 && it < 4000000; ++it) {
        long d = 0;
        if (U(rng) < 0.6) {
            // swap the banks of two variables of the same capacity class (both full-round or both last-round)
            const int a1 = (int)(rng() % na), a2 = (int)(rng() % na);
            if ((a1 < n_fullv) != (a2 < n_fullv)) continue;
            const int v = act_list[a1], w = act_list[a2], bv = bank[v], bw = bank[w];
            if (bv == bw) continue;
            for (int r : var_rows[v]) { d += bump(grp[r], bv, -1); d += bump(grp[r], bw, +1); }
            for (int r : var_rows[w]) { d += bump(grp[r], bw, -1); d += bump(grp[r], bv, +1); }
            if (d <= 0 || U(rng) < std::exp(-(double)d / T)) { bank[v] = bw; bank[w] = bv; cost += d; }
            else {
                for (int r : var_rows[v]) { bump(grp[r], bv, +1); bump(grp[r], bw, -1); }
                for (int r : var_rows[w]) { bump(grp[r], bw, +1); bump(grp[r], bv, -1); }
            }
        } else {
            const int r1 = (int)(rng() % L.m), r2 = (int)(rng() % L.m), g1 = grp[r1], g2 = grp[r2];
            if (g1 == g2 || rdeg(r1) != 7 || rdeg(r2) != 7) continue;
            for (int e = L.row_ptr[r1]; e + 1 < L.row_ptr[r1 + 1]; ++e) { d += bump(g1, bank[L.col[e]], -1); d += bump(g2, bank[L.col[e]], +1); }
            for (int e = L.row_ptr[r2]; e + 1 < L.row_ptr[r2 + 1]; ++e) { d += bump(g2, bank[L.col[e]], -1); d += bump(g1, bank[L.col[e]], +1); }
            if (d <= 0 || U(rng) < std::exp(-(double)d / T)) { grp[r1] = g2; grp[r2] = g1; cost += d; }
            else {
                for (int e = L.row_ptr[r1]; e + 1 < L.row_ptr[r1 + 1]; ++e) { bump(g1, bank[L.col[e]], +1); bump(g2, bank[L.col[e]], -1); }
                for (int e = L.row_ptr[r2]; e + 1 < L.row_ptr[r2 + 1]; ++e) { bump(g2, bank[L.col[e]], +1); bump(g1, bank[L.col[e]], -1); }
            }
        }
        T = std::max(0.05, T * 0.99999);
    }
    if (cost > 0) return ULTRA_HIP_ERR_UNSUPPORTED;

    // variable slots: bank b owns lanes b and b + 32 of every round
    std::vector<int> act_of(L.k, -1);
    {
        std::vector<int> used_full(32, 0), used_last(32, 0);
        for (int a = 0; a < na; ++a) {
            const int j = act_list[a], b = bank[j];
            int slot;
            if (a < n_fullv) { const int u = used_full[b]++; slot = (u / 2) * 64 + b + 32 * (u % 2); if (u / 2 >= VR - 1) return ULTRA_HIP_ERR_UNSUPPORTED; }
            else { const int u = used_last[b]++; if (u >= 2) return ULTRA_HIP_ERR_UNSUPPORTED; slot = (VR - 1) * 64 + b + 32 * u; }
            act_of[j] = slot;
            P.act_var[slot] = (uint16_t)j; P.act_deg[slot] = (uint8_t)vdeg(j);
        }
    }
    // row slots: half-wave g = lanes 32 * (g % 2) .. of round g / 2
    std::vector<int> slot_of(L.m);
    {
        std::vector<int> used(NG, 0);
        for (int i = 0; i < L.m; ++i) {
            const int g = grp[i], u = used[g]++;
            if (u >= 32) return ULTRA_HIP_ERR_UNSUPPORTED;
            slot_of[i] = (g / 2) * 64 + (g % 2) * 32 + u;
        }
    }
    for (int sl = 0; sl < RR * 64; ++sl)
        for (int t = 0; t < 6; ++t) { P.row_addr[6 * sl + t] = 0xFFFF; P.row_col[6 * sl + t] = 0xFFFF; }
    std::mt19937 mrng(0xF117E5u);
    std::vector<uint32_t> mask_of(L.m);
    for (int i = 0; i < L.m; ++i) mask_of[i] = (uint32_t)mrng();
    bool identity = true;
    for (int i = 0; i < L.m; ++i) {
        const int sl = slot_of[i];
        P.row_deg[sl] = (uint8_t)rdeg(i); P.row_id[sl] = (uint16_t)i; P.row_mask[sl] = mask_of[i];
        identity = identity && sl == i;
    }
    // slots of the edges: per half-wave a six-edge-colouring of rows x banks
    std::vector<int> q_of_edge(L.edges, -1);                  // position of the edge among its variable's edges (ascending check)
    for (int j = 0; j < L.k; ++j)
        for (int q = 0; q < vdeg(j); ++q) q_of_edge[L.var_edge[L.var_ptr[j] + q]] = q;
    for (int g = 0; g < NG; ++g) {
        std::vector<int> rows_g;
        for (int i = 0; i < L.m; ++i) if (grp[i] == g) rows_g.push_back(i);
        std::vector<std::pair<int, int>> ed;
        std::vector<int> eid;
        for (size_t li = 0; li < rows_g.size(); ++li)
            for (int e = L.row_ptr[rows_g[li]]; e + 1 < L.row_ptr[rows_g[li] + 1]; ++e) { ed.push_back({(int)li, bank[L.col[e]]}); eid.push_back(e); }
        std::vector<int> colour;
        bipartite_edge_colouring((int)rows_g.size(), 32, ed, 6, colour);
        // The kernel expects the edges of a row in slots 0 .. degree-1.  A permutation of the six colours keeps the
        // colouring proper, so the colours a short row uses are renamed to the first ones (one short row per
        // half-wave at most: two would need compatible colour sets).
        {
            int short_li = -1;
            for (size_t li = 0; li < rows_g.size(); ++li)
                if (rdeg(rows_g[li]) != 7) { if (short_li >= 0) return ULTRA_HIP_ERR_UNSUPPORTED; short_li = (int)li; }
            if (short_li >= 0) {
                int perm[6] = {-1, -1, -1, -1, -1, -1}, next = 0;
                for (size_t x = 0; x < ed.size(); ++x) if (ed[x].first == short_li) perm[colour[x]] = next++;
                for (int c = 0; c < 6; ++c) if (perm[c] < 0) perm[c] = next++;
                for (size_t x = 0; x < ed.size(); ++x) colour[x] = perm[colour[x]];
            }
        }
        for (size_t x = 0; x < ed.size(); ++x) {
            const int e = eid[x], i = rows_g[ed[x].first], t = colour[x], j = L.col[e], a = act_of[j], q = q_of_edge[e];
            if (t < 0 || t >= 6 || q < 0) return ULTRA_HIP_ERR_UNSUPPORTED;
            const int addr = ((a / 64) * dmax + q) * 64 + (a % 64);
            P.row_addr[6 * slot_of[i] + t] = (uint16_t)addr;
            P.row_col[6 * slot_of[i] + t] = (uint16_t)j;
            P.act_addr[a * kLdpcPlanDmax + q] = (uint16_t)addr;
            P.act_mask[a] ^= mask_of[i];
        }
    }
    P.msg_words = VR * dmax * 64;
    P.n_active = na; P.dmax = dmax;
    P.row_rounds = RR; P.var_rounds = VR;
    P.var_rounds_full = VR - 1; P.rows_full = n_short == 0 ? 1 : 0;
    P.row_identity = identity ? 1 : 0;
    P.linear = 1;
    for (int r = 0; r < RR; ++r) {
        int mx = 0, mn = 15;
        for (int sl = 64 * r; sl < 64 * r + 64; ++sl) if (P.row_deg[sl]) { mx = std::max(mx, P.row_deg[sl] - 1); mn = std::min(mn, P.row_deg[sl] - 1); }
        P.prof_rmax |= (uint64_t)mx << (4 * r);
        P.prof_rmin |= (uint64_t)mn << (4 * r);
    }
    for (int r = 0; r < VR; ++r) {
        int mx = 0, mn = 15;
        for (int a = 64 * r; a < 64 * r + 64; ++a) { mx = std::max(mx, (int)P.act_deg[a]); mn = std::min(mn, (int)P.act_deg[a]); }
        P.prof_vmax |= (uint64_t)mx << (4 * r);
        P.prof_vmin |= (uint64_t)mn << (4 * r);
    }
    return ULTRA_HIP_OK;
}

// Plan of the totals kernel from the embedded placement (ldpc_placement.h, generated by tools/ldpc_place.cpp): validated
// against the code's graph — slots unique and in range, at most six edges per (row half-wave, variable bank) cell —, then
// every edge gets its gather instruction on the row side by a six-edge-colouring per row half-wave (conflict-free by
// construction); the variable side gathers in ascending check order (residual collisions there cost cycles, never
// correctness: P.extra_cycles).  ULTRA_HIP_ERR_UNSUPPORTED when the code has no placement or it does not fit.
inline int build_ldpc_tplan(const LdpcConst& L, uint32_t rate, LdpcTPlan& P) {
    P = LdpcTPlan{};
    const uint16_t *vs = nullptr, *rs = nullptr;
    int nv = 0, nr = 0;
    unsigned long long want_rprof = 0, want_vprof = 0;       // low-rate codes: the profiles the placement (and the kernel instance) was made for
    if (!(ldpc_placement(rate, &vs, &nv, &rs, &nr) || ldpc_placement_low(rate, &vs, &nv, &rs, &nr, &want_rprof, &want_vprof)) || nv != L.k ||
        nr != L.m)
        return ULTRA_HIP_ERR_UNSUPPORTED;
    int na = 0, dmax = 0;
    for (int j = 0; j < L.k; ++j) { const int d = L.var_ptr[j + 1] - L.var_ptr[j]; if (d > 0) ++na; dmax = std::max(dmax, d); }
    const int VR = (na + 63) / 64, RR = (L.m + 63) / 64;
    if (VR > kTPlanVarRounds || RR > kTPlanRowRounds || dmax > kTPlanDmax || dmax < 1) return ULTRA_HIP_ERR_UNSUPPORTED;
    for (int j = L.k; j < L.n; ++j) if (L.var_ptr[j + 1] - L.var_ptr[j] != 1) return ULTRA_HIP_ERR_UNSUPPORTED;
    P.k = L.k; P.m = L.m; P.n = L.n; P.max_iterations = L.max_iterations; P.decoded_bytes = L.decoded_bytes;
    P.row_rounds = RR; P.var_rounds = VR; P.dmax = dmax;
    // Degree profiles of the placement: S[r] = the most information edges a row of round r has, D[r] = the most edges a
    // variable of round r has.  The regular codes' kernel (ldpc_totals_kernel.h) runs six slots in every row round whatever
    // the rows hold, so their S is 6 by definition (R2/3 has one row of four).
    int S[kTPlanRowRounds] = {0}, Dv[kTPlanVarRounds] = {0};
    for (int i = 0; i < L.m; ++i) { if (rs[i] >= RR * 64) return ULTRA_HIP_ERR_UNSUPPORTED; S[rs[i] / 64] = std::max(S[rs[i] / 64], L.row_ptr[i + 1] - L.row_ptr[i] - 1); }
    for (int j = 0; j < L.k; ++j) { if (vs[j] == 0xFFFF) continue; if (vs[j] >= VR * 64) return ULTRA_HIP_ERR_UNSUPPORTED; Dv[vs[j] / 64] = std::max(Dv[vs[j] / 64], L.var_ptr[j + 1] - L.var_ptr[j]); }
    if (want_rprof == 0) { for (int r = 0; r < RR; ++r) S[r] = 6; for (int r = 0; r < VR; ++r) Dv[r] = dmax; }
    else {      // the instance's profile must cover what the placement put into each round (it may have slack)
        for (int r = 0; r < RR; ++r) { const int w = (int)((want_rprof >> (4 * r)) & 15ull); if (w < S[r]) return ULTRA_HIP_ERR_UNSUPPORTED; S[r] = w; }
        for (int r = 0; r < VR; ++r) { const int w = (int)((want_vprof >> (4 * r)) & 15ull); if (w < Dv[r]) return ULTRA_HIP_ERR_UNSUPPORTED; Dv[r] = w; }
    }
    P.row_prof = 0; P.var_prof = 0; P.plane_base[0] = 0;
    for (int r = 0; r < RR; ++r) { if (S[r] < 1 || S[r] > 6) return ULTRA_HIP_ERR_UNSUPPORTED; P.row_prof |= (uint64_t)S[r] << (4 * r); P.plane_base[r + 1] = P.plane_base[r] + S[r]; }
    for (int r = 0; r < VR; ++r) { if (Dv[r] < 1 || Dv[r] > 15) return ULTRA_HIP_ERR_UNSUPPORTED; P.var_prof |= (uint64_t)Dv[r] << (4 * r); }
    P.n_planes = P.plane_base[RR];
    // pads: 32 words each (one per bank), so that a pad read can sit in a bank the instruction's real reads leave free
    P.t_pad = VR * 256; P.r_base = P.t_pad + 128; P.r_pad = P.r_base + P.n_planes * 256; P.stage_v = P.r_pad + 128;
    P.stage_p = P.stage_v + VR * 256; P.lds_bytes = P.stage_v + ldpc_stage_bytes(VR, RR);
    for (auto& x : P.row_check) x = 0xFFFF;
    for (auto& x : P.var_id) x = 0xFFFF;
    for (auto& x : P.var_slot_of) x = 0xFFFF;
    for (auto& x : P.row_taddr) x = (uint16_t)P.t_pad;
    for (auto& x : P.var_caddr) x = (uint16_t)P.r_pad;
    // the information bits without any check are the LAST ones of every code the reference builds (buildMatrix connects the
    // first 3 m / ... variables): the decoder's output step relies on that split (ldpc_totals_kernel.h), so it is checked
    P.n_checked = L.k;
    for (int j = L.k - 1; j >= 0 && L.var_ptr[j + 1] - L.var_ptr[j] == 0; --j) P.n_checked = j;
    for (int j = 0; j < P.n_checked; ++j) if (L.var_ptr[j + 1] - L.var_ptr[j] == 0) return ULTRA_HIP_ERR_UNSUPPORTED;
    for (int j = 0; j < L.k; ++j) {
        const int d = L.var_ptr[j + 1] - L.var_ptr[j];
        if ((d > 0) != (vs[j] != 0xFFFF)) return ULTRA_HIP_ERR_UNSUPPORTED;
        if (d == 0) continue;
        if (vs[j] >= VR * 64 || P.var_id[vs[j]] != 0xFFFF) return ULTRA_HIP_ERR_UNSUPPORTED;
        P.var_id[vs[j]] = (uint16_t)j; P.var_slot_of[j] = vs[j];
    }
    for (int i = 0; i < L.m; ++i) {
        const int d = L.row_ptr[i + 1] - L.row_ptr[i];
        if (d < 2 || d > 7 || L.col[L.row_ptr[i + 1] - 1] != L.k + i) return ULTRA_HIP_ERR_UNSUPPORTED;
        if (rs[i] >= RR * 64 || P.row_check[rs[i]] != 0xFFFF || d - 1 > S[rs[i] / 64]) return ULTRA_HIP_ERR_UNSUPPORTED;
        P.row_check[rs[i]] = (uint16_t)i;
    }
    // row side: per half-wave a six-edge-colouring of rows x variable banks
    std::vector<int> slot_t(L.edges, -1);
    for (int g = 0; g < 2 * RR; ++g) {
        std::vector<int> rows_g;
        for (int l = 0; l < 32; ++l) if (P.row_check[g * 32 + l] != 0xFFFF) rows_g.push_back(P.row_check[g * 32 + l]);
        std::vector<std::pair<int, int>> ed; std::vector<int> eid; int cell[32] = {0};
        for (size_t li = 0; li < rows_g.size(); ++li)
            for (int e = L.row_ptr[rows_g[li]]; e + 1 < L.row_ptr[rows_g[li] + 1]; ++e) {
                const int b = vs[L.col[e]] % 32;
                if (++cell[b] > S[g / 2]) return ULTRA_HIP_ERR_UNSUPPORTED;
                ed.push_back({(int)li, b}); eid.push_back(e);
            }
        std::vector<int> colour;
        bipartite_edge_colouring((int)rows_g.size(), 32, ed, S[g / 2], colour);
        for (size_t x = 0; x < ed.size(); ++x) { if (colour[x] < 0 || colour[x] >= S[g / 2]) return ULTRA_HIP_ERR_UNSUPPORTED; slot_t[eid[x]] = colour[x]; }
    }
    std::vector<int> layer(2 * VR * dmax * 32, 0);
    for (int i = 0; i < L.m; ++i)
        for (int e = L.row_ptr[i]; e + 1 < L.row_ptr[i + 1]; ++e) {
            const int j = L.col[e], t = slot_t[e], rsl = rs[i], vsl = vs[j];
            int q = -1;
            for (int x = L.var_ptr[j]; x < L.var_ptr[j + 1]; ++x) if (L.var_edge[x] == e) q = x - L.var_ptr[j];
            if (q < 0 || t < 0) return ULTRA_HIP_ERR_UNSUPPORTED;
            P.row_taddr[rsl * 6 + t] = (uint16_t)(vsl * 4);
            if (q >= Dv[vsl / 64]) return ULTRA_HIP_ERR_UNSUPPORTED;
            P.var_caddr[vsl * kTPlanDmax + q] = (uint16_t)(P.r_base + ((P.plane_base[rsl / 64] + t) * 64 + rsl % 64) * 4);
            layer[((vsl / 32) * dmax + q) * 32 + rsl % 32]++;
        }
    for (int hq = 0; hq < 2 * VR * dmax; ++hq) { int mx = 1; for (int u = 0; u < 32; ++u) mx = std::max(mx, layer[hq * 32 + u]); P.extra_cycles += mx - 1; }
    // Every lane runs every gather (ldpc_totals_kernel.h has no divergent branches), so the operands of lanes without a
    // row / variable, and of missing edges, must not cost LDS cycles either: an idle lane repeats the address of an
    // active lane of its half-wave (identical addresses broadcast), a missing edge reads the pad word of a bank the
    // half-wave's real reads leave free.
    auto fix_half = [&](uint16_t* addr, int n_lanes_base, int stride, int pad_base, auto is_active, auto is_real) {
        bool used[32] = {false};
        int donor = -1;
        for (int l = 0; l < 32; ++l) {
            const int lane = n_lanes_base + l;
            if (is_active(lane) && is_real(lane)) { used[(addr[lane * stride] / 4) % 32] = true; if (donor < 0) donor = lane; }
        }
        for (int l = 0; l < 32; ++l) {
            const int lane = n_lanes_base + l;
            if (is_active(lane) && is_real(lane)) continue;
            if (!is_active(lane) && donor >= 0) { addr[lane * stride] = addr[donor * stride]; continue; }
            int b = 0; while (b < 32 && used[b]) ++b;          // a pad read: free bank (or bank 0 if the half is full)
            if (b == 32) b = 0;
            used[b] = true;
            addr[lane * stride] = (uint16_t)(pad_base + 4 * b);
        }
    };
    for (int g = 0; g < 2 * RR; ++g)
        for (int t = 0; t < 6; ++t)
            fix_half(P.row_taddr + t, g * 32, 6, P.t_pad, [&](int lane) { return P.row_check[lane] != 0xFFFF; },
                     [&](int lane) { return P.row_taddr[lane * 6 + t] < P.t_pad; });
    for (int h = 0; h < 2 * VR; ++h)
        for (int q = 0; q < dmax; ++q)
            fix_half(P.var_caddr + q, h * 32, kTPlanDmax, P.r_pad, [&](int lane) { return P.var_id[lane] != 0xFFFF; },
                     [&](int lane) { return P.var_caddr[lane * kTPlanDmax + q] < P.r_pad; });
    P.valid = 1;
    return ULTRA_HIP_OK;
}

inline int build_ldpc_plan(const LdpcConst& L, LdpcPlan& P, bool allow_linear = true) {
    if (allow_linear && build_ldpc_plan_linear(L, P) == ULTRA_HIP_OK) return ULTRA_HIP_OK;
    P = LdpcPlan{};
    P.k = L.k; P.m = L.m; P.n = L.n; P.edges = L.edges; P.max_iterations = L.max_iterations;
    P.decoded_bytes = L.decoded_bytes;
    if (L.m > kLdpcPlanMaxRows) return ULTRA_HIP_ERR_UNSUPPORTED;
    // slots: rows by information degree, highest first (stable: equal degrees keep their order)
    std::vector<int> slot_row(L.m), slot_of(L.m);
    for (int i = 0; i < L.m; ++i) slot_row[i] = i;
    std::stable_sort(slot_row.begin(), slot_row.end(), [&](int a, int b) {
        return (L.row_ptr[a + 1] - L.row_ptr[a]) > (L.row_ptr[b + 1] - L.row_ptr[b]);
    });
    bool rows_full = true, identity = true;
    for (int sl = 0; sl < L.m; ++sl) {
        const int i = slot_row[sl];
        slot_of[i] = sl;
        identity = identity && i == sl;
        const int deg = L.row_ptr[i + 1] - L.row_ptr[i];
        if (deg < 2 || deg > 7) return ULTRA_HIP_ERR_UNSUPPORTED;          // 1..6 info bits + the parity bit
        if (L.col[L.row_ptr[i + 1] - 1] != L.k + i) return ULTRA_HIP_ERR_UNSUPPORTED;   // identity part last
        P.row_deg[sl] = (uint8_t)deg;
        P.row_id[sl] = (uint16_t)i;
        rows_full = rows_full && deg == 7;
        for (int t = 0; t < 6; ++t) { P.row_addr[6 * sl + t] = 0xFFFF; P.row_col[6 * sl + t] = 0xFFFF; }
    }
    P.row_identity = identity ? 1 : 0;
    for (int j = L.k; j < L.n; ++j)
        if (L.var_ptr[j + 1] - L.var_ptr[j] != 1) return ULTRA_HIP_ERR_UNSUPPORTED;
    int dmax = 0;
    for (int j = 0; j < L.k; ++j) dmax = std::max(dmax, (int)(L.var_ptr[j + 1] - L.var_ptr[j]));
    if (dmax > kLdpcPlanDmax) return ULTRA_HIP_ERR_UNSUPPORTED;

    // active variables by degree, highest first
    std::vector<int> act_of(L.k, -1), act_list;
    for (int j = 0; j < L.k; ++j)
        if (L.var_ptr[j + 1] - L.var_ptr[j] > 0) act_list.push_back(j);
    std::stable_sort(act_list.begin(), act_list.end(), [&](int a, int b) {
        return (L.var_ptr[a + 1] - L.var_ptr[a]) > (L.var_ptr[b + 1] - L.var_ptr[b]);
    });
    if ((int)act_list.size() > kLdpcPlanMaxActive) return ULTRA_HIP_ERR_UNSUPPORTED;
    int na = 0, n_full = 0;
    for (int j : act_list) {
        const int deg = L.var_ptr[j + 1] - L.var_ptr[j];
        act_of[j] = na; P.act_var[na] = (uint16_t)j; P.act_deg[na] = (uint8_t)deg;
        ++na;
        if (deg == dmax) ++n_full;
    }

    // syndrome-filter masks
    std::mt19937 mrng(0xF117E5u);
    for (int i = 0; i < L.m; ++i) P.row_mask[i] = (uint32_t)mrng();

    // information edges with the wave instruction (group) that touches them in each step
    struct Edge { int row, t, act, q; };
    std::vector<Edge> info;
    std::vector<std::pair<int, int>> groups;
    for (int j = 0; j < L.k; ++j) {
        const int a = act_of[j];
        for (int q = 0; q < L.var_ptr[j + 1] - L.var_ptr[j]; ++q) {
            const int e = L.var_edge[L.var_ptr[j] + q];                     // ascending check order
            int check = 0;
            while (!(L.row_ptr[check] <= e && e < L.row_ptr[check + 1])) ++check;
            const int t = e - L.row_ptr[check];
            const int row = slot_of[check];                                 // the row's slot
            info.push_back({row, t, a, q});
            const int cgroup = ((row / 64) * 6 + t) * 2 + ((row % 64) / 32);         // check-step half-wave
            const int vgroup = ((a / 64) * kLdpcPlanDmax + q) * 2 + ((a % 64) / 32);  // variable-step half-wave
            groups.push_back({cgroup, vgroup});
            P.row_col[6 * row + t] = (uint16_t)j;
            P.act_mask[a] ^= P.row_mask[row];
        }
    }
    const int n_cg = ((L.m + 63) / 64) * 6 * 2, n_vg = ((na + 63) / 64) * kLdpcPlanDmax * 2;
    std::vector<int> colour;
    bipartite_edge_colouring(n_cg, n_vg, groups, 32, colour);
    int per_colour[32] = {0};
    int words = 0;
    for (size_t e = 0; e < info.size(); ++e) {
        const int addr = 32 * per_colour[colour[e]]++ + colour[e];          // bank = colour
        words = std::max(words, addr + 1);
        P.row_addr[6 * info[e].row + info[e].t] = (uint16_t)addr;
        P.act_addr[info[e].act * kLdpcPlanDmax + info[e].q] = (uint16_t)addr;
    }
    P.msg_words = (words + 31) & ~31;
    P.n_active = na; P.dmax = dmax;
    P.row_rounds = (L.m + 63) / 64;
    P.var_rounds = (na + 63) / 64;
    P.var_rounds_full = n_full / 64;
    P.rows_full = rows_full ? 1 : 0;
    if (P.row_rounds > 16 || P.var_rounds > 16) return ULTRA_HIP_ERR_UNSUPPORTED;
    for (int r = 0; r < P.row_rounds; ++r) {                                // degree profiles, 4 bits per round
        int mx = 0, mn = 15;
        for (int sl = 64 * r; sl < std::min(L.m, 64 * r + 64); ++sl) { mx = std::max(mx, P.row_deg[sl] - 1); mn = std::min(mn, P.row_deg[sl] - 1); }
        P.prof_rmax |= (uint64_t)mx << (4 * r);
        P.prof_rmin |= (uint64_t)mn << (4 * r);
    }
    for (int r = 0; r < P.var_rounds; ++r) {
        int mx = 0, mn = 15;
        for (int a = 64 * r; a < 64 * r + 64; ++a) { const int d = (a < na) ? P.act_deg[a] : 0; mx = std::max(mx, d); mn = std::min(mn, d); }
        P.prof_vmax |= (uint64_t)mx << (4 * r);
        P.prof_vmin |= (uint64_t)mn << (4 * r);
    }
    return ULTRA_HIP_OK;
}

// ---- demodulator constants ------------------------------------------------
inline int build_demod(const ultra_hip_config& c, DemodConst& D, std::vector<c32>& nco,
                       std::vector<c32>& twiddle) {
    ultra_hip_geometry g;
    int rc = fill_geometry(c, g);
    if (rc != ULTRA_HIP_OK) return rc;
    D = DemodConst{};
    D.fft = (int)c.fft_size;
    D.log2_fft = 0;
    while ((1u << D.log2_fft) < c.fft_size) ++D.log2_fft;
    D.cp = (int)g.cp_len;
    D.sym_len = (int)g.symbol_samples;
    D.n_train = (c.entry == ULTRA_ENTRY_PRESYNCED) ? (int)c.training_symbols : 0;
    D.n_data_sym = (int)c.n_data_symbols;
    D.n_carriers = (int)c.num_carriers;
    D.modulation = (int)c.modulation;
    D.bits = (int)g.bits_per_carrier;
    D.differential = (c.modulation == ULTRA_MOD_DBPSK || c.modulation == ULTRA_MOD_DQPSK ||
                      c.modulation == ULTRA_MOD_D8PSK);
    D.presynced = (c.entry == ULTRA_ENTRY_PRESYNCED);
    D.adaptive_eq = (c.adaptive_eq_enabled && !D.differential) ? (c.adaptive_eq_use_rls ? 2 : 1) : 0;
    D.decision_directed = c.decision_directed ? 1 : 0;
    D.lms_mu = c.lms_mu;
    D.rls_lambda = c.rls_lambda;
    D.fade_k = 0.1f / (float)g.n_data_carriers;
    D.frame_samples = (int)g.frame_samples;
    D.ce_margin = ce_margin(c.modulation);
    D.sample_rate = (float)c.sample_rate;
    D.symbol_duration = static_cast<float>(g.symbol_samples) / static_cast<float>(c.sample_rate);
    D.max_timing = 50.0f * (c.fft_size / 512.0f);
    D.fft_f = (float)c.fft_size;
    D.two_pi_symbol_duration = (2.0 * M_PI) * (double)D.symbol_duration;

    // carrier layout
    const int neg_limit = (int)c.num_carriers / 2, pos_limit = ((int)c.num_carriers + 1) / 2;
    std::vector<bool> pilot_pos;   // "pilot position" irrespective of use_pilots (interp table quirk)
    int slot = 0, count = 0;
    for (int i = -neg_limit; i <= pos_limit; ++i) {
        if (i == 0) continue;
        int bin = (int)((i + c.fft_size) % c.fft_size);
        D.bin[slot] = (int16_t)bin;
        D.k_of[slot] = (int16_t)((bin > (int)c.fft_size / 2) ? bin - (int)c.fft_size : bin);
        bool at_pilot_pos = (count % c.pilot_spacing == 0);
        pilot_pos.push_back(at_pilot_pos);
        if (c.use_pilots && at_pilot_pos) D.pilot_slot[D.n_pilot++] = (int16_t)slot;
        else D.data_slot[D.n_data++] = (int16_t)slot;
        ++slot; ++count;
    }
    D.llrs_per_symbol = D.n_data * D.bits;
    D.llrs_per_frame = D.llrs_per_symbol * D.n_data_sym;
    // the transform keeps bins [0, fq_half) and [fft - fq_half, fft): 32 each when every carrier lies within +-31 (30 and 59
    // carriers do), 64 each otherwise — half the row, half the traffic of three kernels
    D.fq_half = (neg_limit <= 31 && pos_limit <= 31) ? 32 : 64;
    {
        // row positions: pilots, then data carriers, then the unused bins (DemodConst::fq_pos)
        const int rows = 2 * D.fq_half;
        auto natural = [&](int bin) { return (bin < D.fq_half) ? bin : D.fq_half + (bin - (D.fft - D.fq_half)); };
        std::vector<int> pos(rows, -1);
        int next = 0;
        for (int i = 0; i < D.n_pilot; ++i) pos[natural(D.bin[D.pilot_slot[i]])] = next++;
        for (int i = 0; i < D.n_data; ++i) pos[natural(D.bin[D.data_slot[i]])] = next++;
        for (int s = 0; s < rows; ++s) if (pos[s] < 0) pos[s] = next++;
        for (int s = 0; s < 2 * kMaxCarriers; ++s) D.fq_pos[s] = (uint8_t)((s < rows) ? pos[s] : 0);
    }

    // Zadoff-Chu (u = 1) and BPSK pilots
    const size_t N = c.num_carriers, u = 1;
    for (size_t n = 0; n < N; ++n) {
        // the reference's expression is double throughout (M_PI), narrowed once
        const double zc = (((-M_PI * (double)u) * (double)n) * (double)(n + 1)) / (double)N;
        const float phase = (float)zc;
        D.sync_seq[n] = c32{cosf(phase), sinf(phase)};
    }
    std::mt19937 prng(0x50494C54u);   // "PILT"
    for (int i = 0; i < D.n_pilot; ++i) D.pilot_seq[i] = (prng() & 1) ? c32{1.0f, 0.0f} : c32{-1.0f, 0.0f};

    // interpolation table over non-pilot *positions*
    for (int ci = 0; ci < slot; ++ci) {
        if (pilot_pos[ci]) continue;
        int lo = -1, hi = -1;
        for (int j = ci - 1; j >= 0; --j) if (pilot_pos[j]) { lo = j; break; }
        for (int j = ci + 1; j < slot; ++j) if (pilot_pos[j]) { hi = j; break; }
        float alpha = 0.5f;
        if (lo >= 0 && hi >= 0) {
            float total_dist = (float)(hi - lo);
            alpha = (total_dist > 0) ? (float)(ci - lo) / total_dist : 0.5f;
        }
        int q = D.n_interp++;
        D.interp_slot[q] = (int16_t)ci; D.interp_lo[q] = (int16_t)lo; D.interp_hi[q] = (int16_t)hi;
        D.interp_alpha[q] = alpha;
    }

    // NCO sequence after reset() for every sample of a frame
    const size_t total = (size_t)g.frame_samples;
    nco.resize(total);
    {
        // f32 phase accumulator; increment, wrap compare and wrap subtract are
        // evaluated in f64 (2.0f * M_PI is a double) and narrowed on store
        const double two_pi = 2.0 * M_PI;
        const float step = (float)((two_pi * (double)(float)c.center_freq) / (double)(float)c.sample_rate);
        float acc = 0.0f;
        for (size_t i = 0; i < total; ++i) {
            nco[i] = c32{cosf(acc), sinf(acc)};
            acc += step;
            if ((double)acc > two_pi) acc = (float)((double)acc - two_pi);
            if (acc < 0.0f) acc = (float)((double)acc + two_pi);
        }
        D.mixer_phase_end = acc;
    }
    // radix-2 twiddles
    twiddle.resize(c.fft_size / 2);
    for (size_t k = 0; k < c.fft_size / 2; ++k) {
        const float angle = (float)(((-2.0 * M_PI) * (double)k) / (double)c.fft_size);
        twiddle[k] = c32{cosf(angle), sinf(angle)};
    }
    // what the transform's first three stages rely on (demod_kernel.h, group A): the real part of twiddle[0] is exactly 1,
    // the imaginary part of twiddle[N/4] exactly -1 (true for any libm: cos(0) and sin of the float next to -pi/2)
    if (twiddle[0].re != 1.0f || twiddle[c.fft_size / 4].im != -1.0f) return ULTRA_HIP_ERR_UNSUPPORTED;
    return ULTRA_HIP_OK;
}

// LTS passband templates of the demodulator constructor (src/ofdm/demodulator.cpp:100-133): the LTS
// symbol (Zadoff-Chu on the data carriers, BPSK pilots) through the reference's inverse radix-2 FFT
// (src/dsp/fft.cpp:89-121), cyclic prefix prepended, multiplied by a fresh NCO (= the first entries of
// the frame's oscillator table), real and imaginary parts kept.  energy_ref is the serial float sum
// refineLTSTiming recomputes on every call (src/ofdm/ofdm_sync.cpp:405-411).
inline void build_lts_templates(const ultra_hip_config& c, const DemodConst& D, const std::vector<c32>& nco,
                                const std::vector<c32>& twiddle, std::vector<float>& lts_I,
                                std::vector<float>& lts_Q, float& energy_ref) {
    const size_t N = c.fft_size, cp = (size_t)D.cp;
    auto cmulh = [](c32 x, c32 y) {
        const float ac = x.re * y.re, bd = x.im * y.im, ad = x.re * y.im, bc = x.im * y.re;
        return c32{ac - bd, ad + bc};
    };
    std::vector<c32> d(N, c32{0.0f, 0.0f});
    for (int i = 0; i < D.n_data; ++i) d[(size_t)D.bin[D.data_slot[i]]] = D.sync_seq[(size_t)i % c.num_carriers];
    for (int i = 0; i < D.n_pilot; ++i) d[(size_t)D.bin[D.pilot_slot[i]]] = D.pilot_seq[i];
    // fft_impl(inverse = true)
    size_t j = 0;
    for (size_t i = 0; i + 1 < N; ++i) {
        if (i < j) std::swap(d[i], d[j]);
        size_t k = N / 2;
        while (k <= j) { j -= k; k /= 2; }
        j += k;
    }
    for (size_t len = 2; len <= N; len *= 2) {
        const size_t half = len / 2, step = N / len;
        for (size_t i = 0; i < N; i += len)
            for (size_t k = 0; k < half; ++k) {
                c32 w = twiddle[k * step];
                w.im = -w.im;
                const c32 t = cmulh(w, d[i + k + half]);
                const c32 a = d[i + k];
                d[i + k + half] = c32{a.re - t.re, a.im - t.im};
                d[i + k] = c32{a.re + t.re, a.im + t.im};
            }
    }
    const float scale = 1.0f / (float)N;
    for (auto& v : d) v = c32{v.re * scale, v.im * scale};
    lts_I.resize(cp + N);
    lts_Q.resize(cp + N);
    for (size_t i = 0; i < cp + N; ++i) {
        const c32 base = (i < cp) ? d[N - cp + i] : d[i - cp];
        const c32 mixed = cmulh(base, nco[i]);
        lts_I[i] = mixed.re;
        lts_Q[i] = mixed.im;
    }
    energy_ref = 0.0f;
    for (size_t i = 0; i < cp + N; ++i) {
        energy_ref += lts_I[i] * lts_I[i];
        energy_ref += lts_Q[i] * lts_Q[i];
    }
    energy_ref *= 0.5f;
}

// sync::ChirpSync::generateTemplate (src/sync/chirp_sync.hpp:687-732) with OFDMChirpWaveform's configuration
// (src/waveform/ofdm_chirp_waveform.cpp:39-49): sin/cos of the up (300 -> 2700 Hz) and down chirp phases,
// 500 ms long, and the serial float sums of sin^2 the reference keeps as template energies.
struct ChirpHostTables {
    std::vector<float> up_sin, up_cos, dn_sin, dn_cos;
    float e_up = 0.0f, e_dn = 0.0f;
    int len = 0, gap = 0, start_extra = 0;
    float cfo_to_samples = 0.0f;
};
inline void build_chirp_templates(uint32_t sample_rate_u, ChirpHostTables& C) {
    const float fs = static_cast<float>(sample_rate_u), f_start = 300.0f, f_end = 2700.0f, duration_ms = 500.0f,
                gap_ms = 100.0f;
    const size_t len = static_cast<size_t>(fs * duration_ms / 1000.0f);
    const float T = duration_ms / 1000.0f;
    const float k = (f_end - f_start) / T;
    C.up_sin.resize(len); C.up_cos.resize(len); C.dn_sin.resize(len); C.dn_cos.resize(len);
    C.e_up = 0.0f; C.e_dn = 0.0f;
    for (size_t i = 0; i < len; ++i) {
        const float t = static_cast<float>(i) / fs;
        const float phase = (float)(2.0f * M_PI * (double)(f_start * t + 0.5f * k * t * t));     // 2.0f * M_PI is a double
        C.up_sin[i] = sinf(phase); C.up_cos[i] = cosf(phase);
        C.e_up += C.up_sin[i] * C.up_sin[i];
    }
    for (size_t i = 0; i < len; ++i) {
        const float t = static_cast<float>(i) / fs;
        const float phase = (float)(2.0f * M_PI * (double)(f_end * t - 0.5f * k * t * t));
        C.dn_sin[i] = sinf(phase); C.dn_cos[i] = cosf(phase);
        C.e_dn += C.dn_sin[i] * C.dn_sin[i];
    }
    C.len = (int)len;
    C.gap = (int)static_cast<size_t>(fs * gap_ms / 1000.0f);
    // OFDMChirpWaveform::detectSync: chirp_samples + size_t(config_.sample_rate * 100.0f / 1000.0f)
    C.start_extra = (int)(len + static_cast<size_t>(sample_rate_u * 100.0f / 1000.0f));
    const float chirp_rate = (f_end - f_start) / T;
    C.cfo_to_samples = fs / chirp_rate;
}

}  // namespace ultra_hip
#endif
