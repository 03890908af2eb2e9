/* Host check of the routing-schedule builder (smash_amd/csrc/sx_plan.cpp), built with -fsanitize=address,undefined by
 * tests/test_plan_sanitized.py: reads a mesh from stdin (nrow ncol group_size tiled r0 r1 c0 c1, then flwdir and active_cell,
 * column-major), builds the schedule and verifies its invariants -- every active cell of the tile appears in exactly one slot,
 * children are contiguous, exactly one stage below their parent and in D8 order, rounds only receive from earlier rounds, a group
 * never exceeds group_size slots -- then prints a one-line summary. */
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../smash_amd/csrc/sx_plan.h"

#define REQUIRE(c, msg) do { if (!(c)) { std::printf("VIOLATION %s (line %d)\n", msg, __LINE__); return 2; } } while (0)

int main() {
    int nrow, ncol, M, tiled, rect[4];
    if (std::scanf("%d %d %d %d %d %d %d %d", &nrow, &ncol, &M, &tiled, rect, rect + 1, rect + 2, rect + 3) != 8) return 1;
    const long n2 = (long)nrow * ncol;
    std::vector<int> fd(n2), act(n2);
    for (long i = 0; i < n2; ++i) if (std::scanf("%d", &fd[i]) != 1) return 1;
    for (long i = 0; i < n2; ++i) if (std::scanf("%d", &act[i]) != 1) return 1;
    SxSchedule s;
    const int rc = sx_build_schedule(nrow, ncol, fd.data(), act.data(), 0, nullptr, M, tiled ? rect : nullptr, s, nullptr);
    if (rc != 0) { std::printf("rc %d: %s\n", rc, s.error.c_str()); return rc == -5 ? 0 : 3; }
    long want = 0;
    for (int c = 0; c < ncol; ++c)
        for (int r = 0; r < nrow; ++r)
            if (act[r + (long)c * nrow] == 1 && (!tiled || (r >= rect[0] && r < rect[1] && c >= rect[2] && c < rect[3]))) ++want;
    REQUIRE(s.n == want, "cell count");
    std::vector<int> seen(s.n, 0), round_of_group(s.ngroups, -1);
    for (int r = 0; r < s.nrounds; ++r)
        for (int g = s.round_group_begin[r]; g < s.round_group_begin[r + 1]; ++g) round_of_group[g] = r;
    for (int g = 0; g < s.ngroups; ++g) {
        const int b = s.g_slot_begin[g], m = s.g_slot_begin[g + 1] - b;
        REQUIRE(m >= 1 && m <= M, "group size");
        int dmax = 0;
        for (int j = 0; j < m; ++j) {
            const int c = s.s_cell[b + j];
            if (c >= 0) { REQUIRE(c < s.n, "cell index"); seen[c]++; }
            else REQUIRE(-1 - c < std::max(s.nxslots, 1), "inlet series");
            dmax = std::max(dmax, s.s_stage[b + j]);
            const int cs = s.s_cstart[b + j], cc = s.s_ccount[b + j];
            REQUIRE(cc >= 0 && cc <= 8 && (cc == 0 || (cs >= 0 && cs + cc <= m)), "child range");
            for (int q = 0; q < cc; ++q) {
                REQUIRE(s.s_parent[b + cs + q] == j, "parent link");
                REQUIRE(s.s_stage[b + cs + q] == s.s_stage[b + j] - 1, "child one stage below");
            }
            const int par = s.s_parent[b + j];
            REQUIRE(par >= -1 && par < m, "parent index");
            if (c < 0) REQUIRE(cc == 0, "inlet has no children");
        }
        REQUIRE(dmax == s.g_dmax[g], "g_dmax");
    }
    for (int k = 0; k < s.n; ++k) REQUIRE(seen[k] == 1, "every cell exactly once");
    for (int x = 0; x < s.nxslots; ++x) {
        const int pg = s.x_prod_group[x], cg = s.x_cons_group[x];
        if (pg >= 0 && cg >= 0) REQUIRE(round_of_group[pg] < round_of_group[cg], "series flows to a later round");
    }
    REQUIRE(s.out_x.size() == s.out_src.size() && s.in_x.size() == s.in_src.size(), "edge lists");
    for (size_t i = 1; i < s.out_src.size(); ++i) REQUIRE(s.out_src[i - 1] < s.out_src[i], "out edges sorted by source");
    for (size_t i = 1; i < s.in_src.size(); ++i) REQUIRE(s.in_src[i - 1] < s.in_src[i], "in edges sorted by source");
    std::printf("ok cells %d rounds %d groups %d slots %d series %d deepest %d out %zu in %zu\n", s.n, s.nrounds, s.ngroups, s.nslots, s.nxslots,
                s.max_stage, s.out_x.size(), s.in_x.size());
    return 0;
}
