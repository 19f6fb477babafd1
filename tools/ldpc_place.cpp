// tools/ldpc_place.cpp — offline placement solver for the "totals" LDPC kernel (csrc/ldpc_totals_kernel.h).
//
// The kernel keeps ONE total per variable in a lane-linear LDS array T[round][lane] (written by the variable's lane,
// gathered by the rows) and the check-to-variable messages in a lane-linear array R[round][slot][lane] (written by the
// row's lane, gathered by the variables).  Both gathers are conflict-free when
//   C1  for every row half-wave G (32 lanes of a row round) and every LDS bank b (= variable lane mod 32) at most six
//       edges lead from G's rows to variables of bank b — then the bipartite multigraph rows x banks has maximum degree
//       six and a proper six-edge-colouring gives every edge its gather instruction (slot t);
//   C2  for every variable half-wave H and every edge rank q (ascending check order, which the reference's summation
//       order fixes) the q-th rows of H's variables sit at 32 distinct positions (= row lane mod 32).
// Variables may sit at any (round, lane) — one per slot —, rows at any (round, lane).  This is a tight design problem
// (R3/4: 972 edges into 1152 (H, q, position) cells); simulated annealing over both sides finds C1 = 0 and a handful of
// residual C2 collisions (each costs one extra LDS cycle per iteration) in a few seconds — too slow for context creation,
// so the result is generated here once and embedded (csrc/ldpc_placement.h); build_ldpc_tplan validates it against
// the code's graph at run time and falls back to the message-passing kernel if it does not fit.
//
//   g++ -O2 -std=c++17 -Iprojectultra_amd/csrc tools/ldpc_place.cpp -o /tmp/ldpc_place && /tmp/ldpc_place > /tmp/p.h && mv /tmp/p.h projectultra_amd/csrc/ldpc_placement.h
#define ULTRA_LDPC_NO_PLACEMENT 1
#include "host_tables.h"
#include <cstdio>
#include <random>
using namespace ultra_hip;

struct Placement { std::vector<uint16_t> var_slot, row_slot; long c2 = 0; int extra = 0; };

static bool place(uint32_t rate, uint32_t seed, long iters, Placement& out) {
    LdpcConst L; build_ldpc(rate, 50, L);
    std::vector<int> act;
    int dmax = 0;
    for (int j = 0; j < L.k; ++j) { const int d = L.var_ptr[j + 1] - L.var_ptr[j]; if (d > 0) act.push_back(j); dmax = std::max(dmax, d); }
    const int VR = ((int)act.size() + 63) / 64, RR = (L.m + 63) / 64, NH = 2 * VR, NG = 2 * RR, D = dmax;
    for (int i = 0; i < L.m; ++i) { const int d = L.row_ptr[i + 1] - L.row_ptr[i]; if (d < 2 || d > 7 || L.col[L.row_ptr[i + 1] - 1] != L.k + i) return false; }
    std::vector<std::vector<int>> vrow(L.k);
    std::vector<std::vector<std::pair<int, int>>> rvar(L.m);
    for (int i = 0; i < L.m; ++i)
        for (int e = L.row_ptr[i]; e + 1 < L.row_ptr[i + 1]; ++e) vrow[L.col[e]].push_back(i);     // ascending check order
    for (int j = 0; j < L.k; ++j) for (int q = 0; q < (int)vrow[j].size(); ++q) rvar[vrow[j][q]].push_back({j, q});
    std::mt19937 rng(seed);
    std::vector<int> bank(L.k, -1), H(L.k, -1), grp(L.m, -1), U(L.m, -1);
    std::vector<int> occV(32 * NH, -1), occR(NG * 32, -1);
    { int n = 0; for (int j : act) { bank[j] = n % 32; H[j] = n / 32; occV[bank[j] * NH + H[j]] = j; ++n; } }
    { for (int i = 0; i < L.m; ++i) { grp[i] = i % NG; U[i] = i / NG; if (U[i] >= 32) return false; occR[grp[i] * 32 + U[i]] = i; } }
    std::vector<int> c2(NH * D * 32, 0), c1(NG * 32, 0);
    auto pen2 = [](int c) { return c > 1 ? c - 1 : 0; };
    auto pen1 = [](int c) { return c > 6 ? 50 * (c - 6) : 0; };
    auto a2 = [&](int h, int q, int u, int s) { int& c = c2[(h * D + q) * 32 + u]; const int b = pen2(c); c += s; return pen2(c) - b; };
    auto a1 = [&](int g, int b, int s) { int& c = c1[g * 32 + b]; const int bf = pen1(c); c += s; return pen1(c) - bf; };
    long cost = 0;
    for (int j : act) for (int q = 0; q < (int)vrow[j].size(); ++q) { cost += a2(H[j], q, U[vrow[j][q]], 1); cost += a1(grp[vrow[j][q]], bank[j], 1); }
    auto var_at = [&](int j, int b, int h, int s) { long d = 0; for (int q = 0; q < (int)vrow[j].size(); ++q) { const int i = vrow[j][q]; d += a2(h, q, U[i], s); d += a1(grp[i], b, s); } return d; };
    auto row_at = [&](int i, int g, int u, int s) { long d = 0; for (auto [j, q] : rvar[i]) { d += a2(H[j], q, u, s); d += a1(g, bank[j], s); } return d; };
    std::uniform_real_distribution<double> R01(0, 1);
    double T = 0.7;
    std::vector<int> bestV, bestB, bestG, bestU; long best = 1L << 60;
    auto snapshot = [&] { bestV = H; bestB = bank; bestG = grp; bestU = U; best = cost; };
    for (long it = 0; cost > 0 && it < iters; ++it) {
        long d = 0;
        if (rng() % 3) {
            const int j = act[rng() % act.size()], b0 = bank[j], h0 = H[j];
            const int b1 = (rng() % 4 == 0) ? (int)(rng() % 32) : b0, h1 = rng() % NH;
            if (b1 == b0 && h1 == h0) continue;
            const int k = occV[b1 * NH + h1];
            d += var_at(j, b0, h0, -1); if (k >= 0) d += var_at(k, b1, h1, -1);
            d += var_at(j, b1, h1, +1); if (k >= 0) d += var_at(k, b0, h0, +1);
            if (d <= 0 || R01(rng) < std::exp(-(double)d / T)) { bank[j] = b1; H[j] = h1; occV[b1 * NH + h1] = j; if (k >= 0) { bank[k] = b0; H[k] = h0; } occV[b0 * NH + h0] = k; cost += d; }
            else { var_at(j, b1, h1, -1); if (k >= 0) var_at(k, b0, h0, -1); var_at(j, b0, h0, +1); if (k >= 0) var_at(k, b1, h1, +1); }
        } else {
            const int i = rng() % L.m, g0 = grp[i], u0 = U[i];
            const int g1 = (rng() % 4 == 0) ? (int)(rng() % NG) : g0, u1 = rng() % 32;
            if (g1 == g0 && u1 == u0) continue;
            const int k = occR[g1 * 32 + u1];
            d += row_at(i, g0, u0, -1); if (k >= 0) d += row_at(k, g1, u1, -1);
            grp[i] = g1; U[i] = u1; if (k >= 0) { grp[k] = g0; U[k] = u0; }
            d += row_at(i, g1, u1, +1); if (k >= 0) d += row_at(k, g0, u0, +1);
            if (d <= 0 || R01(rng) < std::exp(-(double)d / T)) { occR[g1 * 32 + u1] = i; occR[g0 * 32 + u0] = k; cost += d; }
            else { row_at(i, g1, u1, -1); if (k >= 0) row_at(k, g0, u0, -1); grp[i] = g0; U[i] = u0; if (k >= 0) { grp[k] = g1; U[k] = u1; } row_at(i, g0, u0, +1); if (k >= 0) row_at(k, g1, u1, +1); }
        }
        T = std::max(0.10, T * 0.9999998);
        if (cost < best && T < 0.3) snapshot();
    }
    if (cost < best) snapshot();
    // evaluate the best snapshot
    std::vector<int> e2(NH * D * 32, 0), e1(NG * 32, 0);
    for (int j : act) for (int q = 0; q < (int)vrow[j].size(); ++q) { e2[(bestV[j] * D + q) * 32 + bestU[vrow[j][q]]]++; e1[bestG[vrow[j][q]] * 32 + bestB[j]]++; }
    for (int c : e1) if (c > 6) return false;
    out.c2 = 0; out.extra = 0;
    for (int h = 0; h < NH; ++h) for (int q = 0; q < D; ++q) { int mx = 0; for (int u = 0; u < 32; ++u) { const int c = e2[(h * D + q) * 32 + u]; mx = std::max(mx, c); out.c2 += c > 1 ? c - 1 : 0; } out.extra += std::max(0, mx - 1); }
    out.var_slot.assign(L.k, 0xFFFF); out.row_slot.assign(L.m, 0xFFFF);
    for (int j : act) out.var_slot[j] = (uint16_t)((bestV[j] / 2) * 64 + (bestV[j] % 2) * 32 + bestB[j]);
    for (int i = 0; i < L.m; ++i) out.row_slot[i] = (uint16_t)((bestG[i] / 2) * 64 + (bestG[i] % 2) * 32 + bestU[i]);
    return true;
}

int main(int argc, char** argv) {
    const long iters = argc > 1 ? atol(argv[1]) : 60000000;
    const char* names[6] = {"R1_4", "R1_3", "R1_2", "R2_3", "R3_4", "R5_6"};
    std::printf("// ldpc_placement.h — GENERATED by tools/ldpc_place.cpp (simulated annealing, fixed seeds); do not edit.\n"
                "// Slots of the variables and rows of the totals LDPC kernel (csrc/ldpc_totals_kernel.h): var slot = round * 64 + lane\n"
                "// (0xFFFF: the variable has no check), row slot = round * 64 + lane.  Validated against the code's Tanner graph by\n"
                "// build_ldpc_tplan (csrc/host_tables.h) at context creation.\n"
                "#ifndef ULTRA_LDPC_PLACEMENT_H\n#define ULTRA_LDPC_PLACEMENT_H\n#include <stdint.h>\nnamespace ultra_hip {\n");
    for (uint32_t rate : {3u, 4u, 5u}) {
        Placement best; bool have = false;
        for (uint32_t seed = 1; seed <= 3; ++seed) {
            Placement p;
            if (!place(rate, 1000 * rate + seed, iters, p)) continue;
            std::fprintf(stderr, "rate %s seed %u: residual C2 collisions %ld, extra LDS cycles per iteration %d\n", names[rate], seed, p.c2, p.extra);
            if (!have || p.extra < best.extra) { best = p; have = true; }
            if (best.extra == 0) break;
        }
        if (!have) { std::fprintf(stderr, "rate %s: no placement\n", names[rate]); return 1; }
        std::printf("// %s: %d extra LDS cycles per iteration from residual gather collisions of the variable step\n", names[rate], best.extra);
        std::printf("static const uint16_t kPlaceVar_%s[%zu] = {", names[rate], best.var_slot.size());
        for (size_t i = 0; i < best.var_slot.size(); ++i) std::printf("%s%u", i ? "," : "", best.var_slot[i]);
        std::printf("};\nstatic const uint16_t kPlaceRow_%s[%zu] = {", names[rate], best.row_slot.size());
        for (size_t i = 0; i < best.row_slot.size(); ++i) std::printf("%s%u", i ? "," : "", best.row_slot[i]);
        std::printf("};\n");
    }
    std::printf("inline bool ldpc_placement(uint32_t rate, const uint16_t** var_slot, int* n_var, const uint16_t** row_slot, int* n_row) {\n"
                "    switch (rate) {\n"
                "        case 3: *var_slot = kPlaceVar_R2_3; *n_var = (int)(sizeof(kPlaceVar_R2_3) / 2); *row_slot = kPlaceRow_R2_3; *n_row = (int)(sizeof(kPlaceRow_R2_3) / 2); return true;\n"
                "        case 4: *var_slot = kPlaceVar_R3_4; *n_var = (int)(sizeof(kPlaceVar_R3_4) / 2); *row_slot = kPlaceRow_R3_4; *n_row = (int)(sizeof(kPlaceRow_R3_4) / 2); return true;\n"
                "        case 5: *var_slot = kPlaceVar_R5_6; *n_var = (int)(sizeof(kPlaceVar_R5_6) / 2); *row_slot = kPlaceRow_R5_6; *n_row = (int)(sizeof(kPlaceRow_R5_6) / 2); return true;\n"
                "        default: return false;\n    }\n}\n}  // namespace ultra_hip\n#endif\n");
    return 0;
}
