// tools/ldpc_plan_check.cpp — host check of the LDPC execution plan (csrc/host_tables.h):
// LDS addresses unique, and every half-wave instruction of the check step and of the variable
// step touches 32 distinct banks (the edge colouring is proper).
//   g++ -std=c++17 -O1 -Iprojectultra_amd/csrc tools/ldpc_plan_check.cpp -o /tmp/lpc && /tmp/lpc
#include "host_tables.h"
#include <cstdio>
#include <set>
using namespace ultra_hip;
int main() {
    int total_bad = 0;
    for (uint32_t rate = 0; rate < 6; ++rate) {
        LdpcConst L; build_ldpc(rate, 50, L);
        LdpcPlan P; int rc = build_ldpc_plan(L, P);
        std::set<int> used; int bad = 0;
        for (int i = 0; i < P.row_rounds * 64; i++) for (int t = 0; t < 6; t++) { if (!P.row_deg[i]) continue; int a = P.row_addr[6 * i + t]; if (a != 0xFFFF && !used.insert(a).second) bad++; }
        for (int r = 0; r < P.row_rounds; r++) for (int t = 0; t < 6; t++) for (int h = 0; h < 2; h++) {
            std::set<int> banks;
            for (int l = 0; l < 32; l++) { int row = r * 64 + h * 32 + l; if (!P.row_deg[row]) continue; int a = P.row_addr[6 * row + t]; if (a == 0xFFFF) continue; if (!banks.insert(a % 32).second) bad++; }
        }
        for (int r = 0; r < P.var_rounds; r++) for (int q = 0; q < P.dmax; q++) for (int h = 0; h < 2; h++) {
            std::set<int> banks;
            for (int l = 0; l < 32; l++) { int a = r * 64 + h * 32 + l; if (q >= P.act_deg[a]) continue; int ad = P.act_addr[a * kLdpcPlanDmax + q]; if (P.linear && ad != ((a / 64) * P.dmax + q) * 64 + a % 64) bad++; if (!banks.insert(ad % 32).second) bad++; }
        }
        int info_edges = L.edges - L.m;
        printf("rate %u rc %d info_edges %d unique_addr %zu msg_words %d row_rounds %d var_rounds %d (full %d) rows_full %d dmax %d conflicts %d\n",
               rate, rc, info_edges, used.size(), P.msg_words, P.row_rounds, P.var_rounds, P.var_rounds_full, P.rows_full, P.dmax, bad);
        printf("        profile: rmax 0x%llxull rmin 0x%llxull vmax 0x%llxull vmin 0x%llxull row_identity %d\n",
               (unsigned long long)P.prof_rmax, (unsigned long long)P.prof_rmin, (unsigned long long)P.prof_vmax,
               (unsigned long long)P.prof_vmin, P.row_identity);
        // row_id is a permutation of the checks and the slots are sorted by degree
        std::set<int> ids;
        int n_slots = 0, n_vars = 0, var_edges = 0;
        for (int sl = 0; sl < P.row_rounds * 64; sl++) {
            if (!P.row_deg[sl]) continue;
            ++n_slots; ids.insert(P.row_id[sl]);
            if (!P.linear && sl > 0 && P.row_deg[sl] > P.row_deg[sl - 1]) bad++;      // sorted by degree (arbitrary-address plans)
        }
        if ((int)ids.size() != L.m || n_slots != L.m) bad++;
        for (int sl = 0; sl < P.row_rounds * 64; sl++)          // a row's edges sit in slots 0 .. degree-2 (the parity edge is the last)
            for (int t = 0; t < 6; t++) if (P.row_deg[sl] && (P.row_addr[6 * sl + t] != 0xFFFF) != (t < P.row_deg[sl] - 1)) bad++;
        for (int a = 0; a < P.var_rounds * 64; a++) if (P.act_deg[a]) { ++n_vars; var_edges += P.act_deg[a]; }
        if (n_vars != P.n_active || var_edges != info_edges) bad++;
        // every edge is known to both sides under the same address
        { std::multiset<int> ra, va;
          for (int sl = 0; sl < P.row_rounds * 64; sl++) for (int t = 0; t < 6; t++) if (P.row_deg[sl] && P.row_addr[6 * sl + t] != 0xFFFF) ra.insert(P.row_addr[6 * sl + t]);
          for (int a = 0; a < P.var_rounds * 64; a++) for (int q = 0; q < P.act_deg[a]; q++) va.insert(P.act_addr[a * kLdpcPlanDmax + q]);
          if (ra != va) bad++; }
        printf("        linear %d\n", P.linear);
        if (rc != 0 || bad != 0 || (int)used.size() != info_edges) total_bad++;
    }
    return total_bad ? 1 : 0;
}
