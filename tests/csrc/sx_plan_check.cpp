/* Host check of the routing-schedule builder (smash_amd/csrc/sx_plan.cpp), built with -fsanitize=address,undefined by
 * tests/test_plan_sanitized.py: reads a mesh from stdin (nrow ncol group_size tiled r0 r1 c0 c1, then flwdir and active_cell,
 * column-major, optionally the sub-level count), builds the schedule and verifies its invariants -- every active cell of the tile
 * appears in exactly one slot; a child is either in its parent's component (same stage, same wavefront, lower sub-level) or exactly
 * one stage below it; child lists and parent links agree; sub-levels stay below the limit and below the wavefront's count; rounds
 * only receive from earlier rounds; a group never exceeds group_size slots -- then prints a one-line summary. */
#include <algorithm>
#include <climits>
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
    int U = 4;
    if (std::scanf("%d", &U) != 1) U = 4;
    SxSchedule s;
    const int rc = sx_build_schedule(nrow, ncol, fd.data(), act.data(), 0, nullptr, M, tiled ? rect : nullptr, s, nullptr, U | (U << 8));
    if (rc != 0) { std::printf("rc %d: %s\n", rc, s.error.c_str()); return rc == -5 ? 0 : 3; }
    long want = 0;
    for (int c = 0; c < ncol; ++c)
        for (int r = 0; r < nrow; ++r)
            if (act[r + (long)c * nrow] == 1 && (!tiled || (r >= rect[0] && r < rect[1] && c >= rect[2] && c < rect[3]))) ++want;
    REQUIRE(s.n == want, "cell count");
    std::vector<int> seen(s.n, 0), round_of_group(s.ngroups, -1);
    for (int r = 0; r < s.nrounds; ++r)
        for (int g = s.round_group_begin[r]; g < s.round_group_begin[r + 1]; ++g) round_of_group[g] = r;
    long stage_sum = 0;
    for (int g = 0; g < s.ngroups; ++g) {
        const int b = s.g_slot_begin[g], m = s.g_slot_begin[g + 1] - b;
        REQUIRE(m >= 1 && m <= M, "group size");
        int dmax = 0;
        for (int j = 0; j < m; ++j) {
            const int c = s.s_cell[b + j];
            if (c == INT_MIN) { REQUIRE(s.s_ccount[b + j] == 0 && s.s_parent[b + j] == -1, "empty slot carries links"); continue; }
            if (c >= 0) { REQUIRE(c < s.n, "cell index"); seen[c]++; }
            else REQUIRE(-1 - c < std::max(s.nxslots, 1), "inlet series");
            dmax = std::max(dmax, s.s_stage[b + j]);
            REQUIRE(s.s_stage[b + j] >= 0, "stage");
            REQUIRE(s.s_sub[b + j] >= 0 && s.s_sub[b + j] < U && s.s_sub[b + j] < s.s_wsub[b + j], "sub-level range");
            REQUIRE(s.s_wsub[b + j] == s.s_wsub[b + (j / 64) * 64], "wavefront sub-level count is wave-uniform");
            const int cc = s.s_ccount[b + j];
            REQUIRE(cc >= 0 && cc <= 8, "child count");
            int need_sub = 0;
            for (int q = 0; q < 8; ++q) {
                const unsigned wd = (unsigned)s.s_child[(size_t)(b + j) * 4 + q / 2];
                const unsigned e = (wd >> ((q & 1) * 16)) & 0xffffu;
                if (q >= cc) { REQUIRE(e == 0xffffu, "unused child entry"); continue; }
                const int ci = (int)(e & 0x7fffu);
                const bool same = (e & 0x8000u) != 0;
                REQUIRE(ci < m && s.s_cell[b + ci] != INT_MIN, "child index");
                REQUIRE((s.s_parent[b + ci] & 0xffff) == j && s.s_parent[b + ci] >= 0, "parent link");
                REQUIRE(((s.s_parent[b + ci] & 0x40000000) != 0) == same, "component flag agrees on both ends");
                if (same) {
                    REQUIRE(s.s_stage[b + ci] == s.s_stage[b + j], "same component, same stage");
                    REQUIRE(ci / 64 == j / 64, "same component, same wavefront");
                    REQUIRE(s.s_sub[b + ci] < s.s_sub[b + j], "child of the same component sits on a lower sub-level");
                    need_sub = std::max(need_sub, s.s_sub[b + ci] + 1);
                } else {
                    REQUIRE(s.s_stage[b + ci] == s.s_stage[b + j] - 1, "child component exactly one stage below");
                }
            }
            REQUIRE(s.s_sub[b + j] == need_sub, "sub-level is one above the highest child of the component");
            const int par = s.s_parent[b + j];
            REQUIRE(par >= -1 && (par < 0 || (par & 0xffff) < m), "parent index");
            if (c < 0) REQUIRE(cc == 0, "inlet has no children");
        }
        REQUIRE(dmax == s.g_dmax[g], "g_dmax");
        stage_sum += dmax;
    }
    for (int k = 0; k < s.n; ++k) REQUIRE(seen[k] == 1, "every cell exactly once");
    for (int x = 0; x < s.nxslots; ++x) {
        const int pg = s.x_prod_group[x], cg = s.x_cons_group[x];
        if (pg >= 0 && cg >= 0) REQUIRE(round_of_group[pg] < round_of_group[cg], "series flows to a later round");
    }
    // round 4: the series are numbered in the order of the inlets that read them -- walking the slots in order meets the series 0, 1, 2, ...
    // (so the inlets of consecutive slots read one contiguous piece of a row of the exchange array); series without an inlet in this
    // tile (they only leave it) come after all of those.  And the subtrees are packed in that order without giving up the fill.
    {
        int next = 0;
        for (int q = 0; q < s.nslots; ++q) {
            const int c = s.s_cell[q];
            if (c >= 0 || c == INT_MIN) continue;
            REQUIRE(-1 - c == next, "series numbered in the order of their inlets");
            ++next;
        }
        for (int x = next; x < s.nxslots; ++x) REQUIRE(s.x_cons_group[x] < 0, "a series without inlet here is read by another tile only");
        const int g0 = s.round_group_begin[0], g1 = s.round_group_begin[1];
        const long slots0 = s.g_slot_begin[g1] - s.g_slot_begin[g0];
        if (g1 - g0 >= 8) REQUIRE(slots0 >= (long)(0.95 * M) * (g1 - g0 - 1), "groups of round 0 at least 95 % full");
        // the roots of a group publish within a window of a few times their number
        long outs = 0, span = 0;
        for (int g = g0; g < g1; ++g) {
            int lo = INT_MAX, hi = -1, cnt = 0;
            for (int q = s.g_slot_begin[g]; q < s.g_slot_begin[g + 1]; ++q) { const int x = s.s_xout[q]; if (x >= 0 && s.x_cons_group[x] >= 0) { lo = std::min(lo, x); hi = std::max(hi, x); ++cnt; } }
            if (cnt) { outs += cnt; span += hi - lo + 1; }
        }
        if (outs >= 1000) REQUIRE(span <= 4 * outs, "the roots of a round-0 group publish next to each other");
    }
    REQUIRE(s.out_x.size() == s.out_src.size() && s.in_x.size() == s.in_src.size(), "edge lists");
    for (size_t i = 1; i < s.out_src.size(); ++i) REQUIRE(s.out_src[i - 1] < s.out_src[i], "out edges sorted by source");
    for (size_t i = 1; i < s.in_src.size(); ++i) REQUIRE(s.in_src[i - 1] < s.in_src[i], "in edges sorted by source");
    std::printf("ok cells %d rounds %d groups %d slots %d series %d deepest %d out %zu in %zu stagesum %ld\n", s.n, s.nrounds, s.ngroups, s.nslots,
                s.nxslots, s.max_stage, s.out_x.size(), s.in_x.size(), stage_sum);
    return 0;
}
