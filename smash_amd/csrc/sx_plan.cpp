// sx_plan.cpp -- builds the routing schedule (see sx_plan.h).  Host-only C++.
#include "sx_plan.h"

#include <algorithm>
#include <climits>
#include <cstdint>
#include <map>
#include <numeric>

namespace {
// D8: code k = 1..8 = N, NE, E, SE, S, SW, W, NW flows to (row + DROW[k-1], col + DCOL[k-1])
// (reference smash/mesh/mw_meshing.f90:163-164); the solver's upstream test
// (md_routing_operator.f90:29-31,45) is the mirror image: neighbour i drains into me iff its code == i.
const int DROW[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
const int DCOL[8] = {0, 1, 1, 1, 0, -1, -1, -1};
}  // namespace

int sx_build_schedule(int nrow, int ncol, const int* flwdir, const int* active_cell, int ng, const int* gauge_pos,
                      int group_size, const int* rect, SxSchedule& s, const int* own, int sublevels) {
    const int M = group_size;
    if (nrow <= 0 || ncol <= 0 || M < 16) { s.error = "bad sizes"; return -1; }
    const long n2 = (long)nrow * ncol;
    s.nrow = nrow; s.ncol = ncol; s.group_size = M;

    // ---- local active cells in column-major order (temporary index a), parents, children in D8 order ----
    const int r0 = rect ? rect[0] : 0, r1 = rect ? rect[1] : nrow, c0 = rect ? rect[2] : 0, c1 = rect ? rect[3] : ncol;
    auto inside = [&](int row, int col) {
        if (own) return own[row + (long)col * nrow] == 1;
        return row >= r0 && row < r1 && col >= c0 && col < c1;
    };
    const bool part = rect || own;
    std::vector<int> a_of_flat(n2, -1), flat_of_a;
    flat_of_a.reserve((size_t)(r1 - r0) * (c1 - c0));
    for (long c = 0; c < n2; ++c)
        if (active_cell[c] == 1 && inside((int)(c % nrow), (int)(c / nrow))) { a_of_flat[c] = (int)flat_of_a.size(); flat_of_a.push_back((int)c); }
    const int n = (int)flat_of_a.size();
    s.n = n;
    if (n == 0) { s.error = "no active cell"; return -1; }
    // cells of other tiles draining into a local cell: virtual nodes n .. n+nr-1 (always inlets)
    std::vector<int> rflat, rparent;
    std::vector<int> parent(n, -1), code(n, 0), remote_parent_flat(n, -1);
    for (int a = 0; a < n; ++a) {
        const int c = flat_of_a[a], row = c % nrow, col = c / nrow, fd = flwdir[c];
        code[a] = fd;
        if (part)
            for (int i = 0; i < 8; ++i) {   // neighbour at -D[i] drains into me iff its code == i+1
                const int rn = row - DROW[i], cn = col - DCOL[i];
                if (rn < 0 || rn >= nrow || cn < 0 || cn >= ncol || inside(rn, cn)) continue;
                const long fn = rn + (long)cn * nrow;
                if (active_cell[fn] == 1 && flwdir[fn] == i + 1) { rflat.push_back((int)fn); rparent.push_back(a); }
            }
        if (fd < 1 || fd > 8) continue;
        const int r2 = row + DROW[fd - 1], c2 = col + DCOL[fd - 1];
        if (r2 < 0 || r2 >= nrow || c2 < 0 || c2 >= ncol) continue;
        const long f2 = r2 + (long)c2 * nrow;
        if (active_cell[f2] != 1) continue;            // receiver inactive: catchment outlet
        if (inside(r2, c2)) parent[a] = a_of_flat[f2];
        else remote_parent_flat[a] = (int)f2;          // receiver lives in another tile
    }
    const int nr = (int)rflat.size(), nv = n + nr;
    code.resize(nv);
    for (int q = 0; q < nr; ++q) code[n + q] = flwdir[rflat[q]];
    std::vector<int> nchild(n, 0), cbeg(n + 1, 0);
    for (int a = 0; a < n; ++a) if (parent[a] >= 0) nchild[parent[a]]++;
    for (int q = 0; q < nr; ++q) nchild[rparent[q]]++;
    for (int a = 0; a < n; ++a) cbeg[a + 1] = cbeg[a] + nchild[a];
    std::vector<int> child(cbeg[n]), fill(n, 0);
    for (int a = 0; a < n; ++a) if (parent[a] >= 0) { const int p = parent[a]; child[cbeg[p] + fill[p]++] = a; }
    for (int q = 0; q < nr; ++q) { const int p = rparent[q]; child[cbeg[p] + fill[p]++] = n + q; }
    for (int a = 0; a < n; ++a)   // order 1..8 as md_routing_operator.f90:37-53 sums them
        std::sort(child.begin() + cbeg[a], child.begin() + cbeg[a + 1], [&](int x, int y) { return code[x] < code[y]; });

    // ---- topological order (leaves first); a cycle (pit pair) cannot be scheduled ----
    std::vector<int> topo; topo.reserve(n);
    {
        std::vector<int> indeg(n, 0);
        for (int a = 0; a < n; ++a) if (parent[a] >= 0) indeg[parent[a]]++;
        for (int a = 0; a < n; ++a) if (indeg[a] == 0) topo.push_back(a);
        for (size_t i = 0; i < topo.size(); ++i) {
            const int p = parent[topo[i]];
            if (p >= 0 && --indeg[p] == 0) topo.push_back(p);
        }
        if ((int)topo.size() != n) { s.error = "flow directions contain a cycle over active cells"; return -5; }
    }

    // ---- rounds: repeatedly peel off every maximal subtree whose weight (cells + inlets) fits a group ----
    std::vector<int> round_of(nv, -1), w(nv, 0);
    for (int q = 0; q < nr; ++q) round_of[n + q] = -2;   // remote: never scheduled here, always an inlet
    struct Root { int a; int weight; };
    std::vector<std::vector<Root>> round_roots;                 // round -> its subtree roots (packed into groups further down)
    int remaining = n, round = 0;
    std::vector<int> todo(topo);
    while (remaining > 0) {
        for (int a : todo) {
            long ww = 1;
            for (int j = cbeg[a]; j < cbeg[a + 1]; ++j) ww += (round_of[child[j]] != -1) ? 1 : w[child[j]];
            w[a] = (int)std::min<long>(ww, (long)M + 1);
        }
        std::vector<Root> roots;
        for (int a : todo)
            if (w[a] <= M && (parent[a] < 0 || w[parent[a]] > M)) roots.push_back({a, w[a]});
        if (roots.empty()) { s.error = "schedule made no progress (group_size too small)"; return -1; }
        std::vector<int> next; next.reserve(todo.size());
        for (int a : todo) { if (w[a] <= M) { round_of[a] = round; --remaining; } else next.push_back(a); }
        todo.swap(next);
        round_roots.push_back(std::move(roots));
        ++round;
    }
    s.nrounds = round;

    // ---- group-local layout: components (one wavefront, <= U levels), stages, sub-levels, device cell numbering ----
    // sublevels = levels per super-step in the rounds >= 1 (low byte) | levels in round 0 << 8 (0: one).  Round 0 is wide and bound by
    // HBM: every extra pass through the super-step's body costs it vector time it does not have (measured: 4 levels everywhere make a
    // 1024^2 x 8760 sweep 22 ms slower), the later rounds are bound by the latency of their chain of stages
    const int U_late = std::max(1, std::min(sublevels & 0xff, 16)), U_first = std::max(1, std::min((sublevels >> 8) & 0xff, 16));
    const int WAVE = 64, nwaves = (M + WAVE - 1) / WAVE;
    // the forest of one group in breadth-first order (parents before children), node >= 0 cell a, < 0 inlet of node -1-a, and where
    // every node sits in the group
    struct GroupLayout {
        std::vector<int> q_node, q_par, kid_begin, kids, slot_of, stage, sub, wsub;
        std::vector<char> merged;
        int m = 0, m_slots = 0, dmax = 0;
    };
    auto layout_group = [&](const std::vector<int>& g, int r, int U, GroupLayout& G) -> int {
        std::vector<int>&q_node = G.q_node, &q_par = G.q_par, &kid_begin = G.kid_begin, &kids = G.kids;
        q_node.clear(); q_par.clear(); kid_begin.assign(1, 0); kids.clear();
        for (int a : g) { q_node.push_back(a); q_par.push_back(-1); }
        for (size_t i = 0; i < q_node.size(); ++i) {
            const int node = q_node[i];
            if (node >= 0)
                for (int j = cbeg[node]; j < cbeg[node + 1]; ++j) {
                    const int c = child[j];
                    kids.push_back((int)q_node.size());
                    q_node.push_back((c < n && round_of[c] == r) ? c : -1 - c);
                    q_par.push_back((int)i);
                }
            kid_begin.push_back((int)kids.size());
        }
        const int m = (int)q_node.size();
        if (m > M) { s.error = "internal: group overflow"; return -1; }
        // -- components, bottom-up (children have larger local indices than their parent): a node takes the components of as many
        //    children as fit one wavefront and U levels, the child with the deepest component tree below it first (that is the
        //    path the fill runs along)
        std::vector<int> comp(m), csize(m, 1), cheight(m, 1), cdepth(m, 1);   // comp: representative (the component's top node)
        std::vector<char>& merged = G.merged;                                  // node i sits in its parent's component
        merged.assign(m, 0);
        for (int i = 0; i < m; ++i) comp[i] = i;
        auto build_components = [&](const std::vector<char>& forbid) {
            for (int i = m - 1; i >= 0; --i) {
                csize[i] = 1; cheight[i] = 1; cdepth[i] = 1;
                std::vector<int> ks(kids.begin() + kid_begin[i], kids.begin() + kid_begin[i + 1]);
                std::stable_sort(ks.begin(), ks.end(), [&](int x, int y) { return cdepth[x] != cdepth[y] ? cdepth[x] > cdepth[y] : csize[x] > csize[y]; });
                for (int c : ks) {
                    merged[c] = 0;
                    if (U > 1 && !forbid[c] && csize[i] + csize[c] <= WAVE && cheight[c] + 1 <= U) {
                        merged[c] = 1;
                        csize[i] += csize[c];
                        cheight[i] = std::max(cheight[i], cheight[c] + 1);
                        cdepth[i] = std::max(cdepth[i], cdepth[c]);
                    } else {
                        cdepth[i] = std::max(cdepth[i], cdepth[c] + 1);
                    }
                }
            }
            for (int i = 0; i < m; ++i) comp[i] = (q_par[i] >= 0 && merged[i]) ? comp[q_par[i]] : i;
        };
        // -- pack the components into wavefronts (first fit, largest first); one that fits nowhere is split at its top node (its
        //    merged children become components of their own) and everything is rebuilt: single slots always fit
        std::vector<char> forbid(m, 0);
        std::vector<int> wave_of(m, -1);
        for (;;) {
            build_components(forbid);
            std::vector<int> tops;
            for (int i = 0; i < m; ++i) if (comp[i] == i) tops.push_back(i);
            std::stable_sort(tops.begin(), tops.end(), [&](int x, int y) { return csize[x] > csize[y]; });
            std::vector<int> room(nwaves, WAVE);
            for (int w2 = 0; w2 < nwaves; ++w2) room[w2] = std::max(0, std::min(WAVE, M - w2 * WAVE));   // (M below a wavefront multiple)
            int bad = -1;
            std::vector<int> wtop(m, -1);
            for (int t : tops) {
                int w2 = 0;
                while (w2 < nwaves && room[w2] < csize[t]) ++w2;
                if (w2 == nwaves) { bad = t; break; }
                room[w2] -= csize[t]; wtop[t] = w2;
            }
            if (bad < 0) { for (int i = 0; i < m; ++i) wave_of[i] = wtop[comp[i]]; break; }
            bool any = false;
            for (int j = kid_begin[bad]; j < kid_begin[bad + 1]; ++j) if (merged[kids[j]]) { forbid[kids[j]] = 1; any = true; }
            if (!any) { s.error = "internal: component packing"; return -1; }
        }
        // -- stages top-down over the component tree, sub-levels inside the components
        int dmax = 0;
        for (int i = 0; i < m; ++i) if (q_par[i] < 0) dmax = std::max(dmax, cdepth[i] - 1);
        std::vector<int>&stage = G.stage, &sub = G.sub;
        stage.assign(m, 0); sub.assign(m, 0);
        for (int i = 0; i < m; ++i) stage[i] = q_par[i] < 0 ? dmax : (merged[i] ? stage[q_par[i]] : stage[q_par[i]] - 1);
        for (int i = m - 1; i >= 0; --i)
            for (int j = kid_begin[i]; j < kid_begin[i + 1]; ++j) if (merged[kids[j]]) sub[i] = std::max(sub[i], sub[kids[j]] + 1);
        for (int i = 0; i < m; ++i) if (stage[i] < 0) { s.error = "internal: negative stage"; return -1; }
        // -- slot order: wave by wave, inside a wave in local (breadth-first) order; unused lanes of a wave stay empty only at the end
        std::vector<int> order(m);
        std::vector<int>& slot_of = G.slot_of;
        slot_of.assign(m, 0);
        for (int i = 0; i < m; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return wave_of[x] < wave_of[y]; });
        // waves must start at multiples of 64: pad with nothing by construction only if every earlier wave is full -- otherwise
        // shift: slots are placed at wave * 64 + position, and the group's slot count covers the holes with empty slots
        std::vector<int> fill2(nwaves, 0);
        int m_slots = 0;
        for (int i : order) { slot_of[i] = wave_of[i] * WAVE + fill2[wave_of[i]]++; m_slots = std::max(m_slots, slot_of[i] + 1); }
        if (m_slots > M) { s.error = "internal: group overflow after packing"; return -1; }
        G.wsub.assign(nwaves, 1);
        for (int i = 0; i < m; ++i) G.wsub[wave_of[i]] = std::max(G.wsub[wave_of[i]], sub[i] + 1);
        G.m = m; G.m_slots = m_slots; G.dmax = dmax;
        return 0;
    };

    // ---- groups, last round first.  A subtree hands its root's series to ONE inlet of a group of a later round; the subtrees of a round
    // are packed into groups IN THE ORDER OF THOSE INLETS (next fit), and the series are numbered in the order of their inlets (below).
    // So the inlets of 64 consecutive slots of a consumer read one contiguous piece of a row of the exchange array, and the roots of a
    // producer group -- which feed neighbouring inlets -- still write one: both ends of a series are coalesced (round 4; before, series
    // were numbered in producer order and a consumer's inlets gathered 16-byte pieces from all over a row, which is what kept the
    // transposition of the chained groups' inputs -- sx_kernels.h "Staging rows" -- from paying).  Best-fit packing filled the groups
    // to 99.9 %; next fit in inlet order leaves about one subtree's worth of room per group (average subtree: 17 slots of 512).
    std::vector<std::vector<GroupLayout>> L(s.nrounds);
    std::vector<std::vector<std::vector<int>>> round_groups(s.nrounds);   // round -> group -> list of roots
    std::vector<long long> inlet_key(nv, -1);      // for a node that is an inlet somewhere: (round, group, slot) of that inlet
    for (int r = s.nrounds - 1; r >= 0; --r) {
        const int U = r == 0 ? U_first : U_late;
        std::vector<Root>& roots = round_roots[r];
        // roots whose series goes to another tile or nowhere (catchment outlets) come last, in mesh order
        std::stable_sort(roots.begin(), roots.end(), [&](const Root& x, const Root& y) {
            const long long kx = inlet_key[x.a] < 0 ? LLONG_MAX : inlet_key[x.a], ky = inlet_key[y.a] < 0 ? LLONG_MAX : inlet_key[y.a];
            return kx != ky ? kx < ky : flat_of_a[x.a] < flat_of_a[y.a];
        });
        std::vector<std::vector<int>>& groups = round_groups[r];
        // first fit in inlet order with a bounded look-ahead: a group takes the next subtrees in order, and when one does not fit, the
        // first of the following `ahead` that do; the ones passed over open the next group
        const size_t ahead = 96;
        std::vector<char> used(roots.size(), 0);
        for (size_t head = 0; head < roots.size();) {
            if (used[head]) { ++head; continue; }
            groups.emplace_back();
            int room = M;
            size_t seen = 0;
            for (size_t i = head; i < roots.size() && seen < ahead && room > 0; ++i) {
                if (used[i]) continue;
                ++seen;
                if (roots[i].weight <= room) { used[i] = 1; groups.back().push_back(roots[i].a); room -= roots[i].weight; if (i == head) seen = 0; }
            }
        }
        L[r].resize(groups.size());
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            GroupLayout& G = L[r][gi];
            if (layout_group(groups[gi], r, U, G) != 0) return -1;
            for (int i = 0; i < G.m; ++i)
                if (G.q_node[i] < 0) inlet_key[-1 - G.q_node[i]] = ((long long)r << 44) | ((long long)gi << 20) | (long long)G.slot_of[i];
        }
    }

    // ---- exchange series: one per subtree root that has a receiver in a later round or in another tile, one per cell of another tile that
    // drains into this one -- numbered in the order of the inlets that read them; series that only leave the tile come last
    // (roots without parent are catchment outlets: they publish nothing)
    std::vector<int> xslot_of(nv, -1);
    int nx = 0;
    for (int r = 0; r < s.nrounds; ++r)
        for (const GroupLayout& G : L[r]) {
            std::vector<int> by_slot(G.m_slots, -1);
            for (int i = 0; i < G.m; ++i) by_slot[G.slot_of[i]] = i;
            for (int q = 0; q < G.m_slots; ++q)
                if (by_slot[q] >= 0 && G.q_node[by_slot[q]] < 0) xslot_of[-1 - G.q_node[by_slot[q]]] = nx++;
        }
    for (int r = 0; r < s.nrounds; ++r)
        for (auto& g : round_groups[r])
            for (int a : g) if (xslot_of[a] < 0 && remote_parent_flat[a] >= 0) xslot_of[a] = nx++;
    for (int q = 0; q < nr; ++q) if (xslot_of[n + q] < 0) { s.error = "internal: a boundary inlet was not laid out"; return -1; }
    s.nxslots = nx;
    {   // boundary series, sorted by source cell so both sides of a tile edge enumerate them identically
        std::vector<int> o;
        for (int a = 0; a < n; ++a) if (remote_parent_flat[a] >= 0) o.push_back(a);
        std::sort(o.begin(), o.end(), [&](int x, int y) { return flat_of_a[x] < flat_of_a[y]; });
        for (int a : o) { s.out_x.push_back(xslot_of[a]); s.out_src.push_back(flat_of_a[a]); s.out_dst.push_back(remote_parent_flat[a]); }
        std::vector<int> in(nr);
        std::iota(in.begin(), in.end(), 0);
        std::sort(in.begin(), in.end(), [&](int x, int y) { return rflat[x] < rflat[y]; });
        for (int q : in) { s.in_x.push_back(xslot_of[n + q]); s.in_src.push_back(rflat[q]); s.in_dst.push_back(flat_of_a[rparent[q]]); }
    }

    // ---- the schedule's arrays, rounds ascending; cells are numbered in slot order
    s.round_group_begin.assign(1, 0);
    s.g_slot_begin.assign(1, 0);
    std::vector<int> k_of_a(n, -1);
    int knext = 0;
    s.max_stage = 0;
    s.x_prod_group.assign(std::max(nx, 1), -1);
    s.x_cons_group.assign(std::max(nx, 1), -1);
    for (int r = 0; r < s.nrounds; ++r) {
        for (const GroupLayout& G : L[r]) {
            const int m = G.m, m_slots = G.m_slots, dmax = G.dmax;
            const std::vector<int>&q_node = G.q_node, &q_par = G.q_par, &kid_begin = G.kid_begin, &kids = G.kids, &slot_of = G.slot_of,
                                  &stage = G.stage, &sub = G.sub;
            const std::vector<char>& merged = G.merged;
            s.g_dmax.push_back(dmax);
            s.max_stage = std::max(s.max_stage, dmax);
            const int gi = (int)s.g_slot_begin.size() - 1, sbase = (int)s.s_cell.size();
            // empty slots (holes of partly filled waves): an inlet of no series is not expressible, so they are marked by cell = INT_MIN
            s.s_cell.resize(sbase + m_slots, INT32_MIN);
            s.s_stage.resize(sbase + m_slots, 0); s.s_sub.resize(sbase + m_slots, 0); s.s_wsub.resize(sbase + m_slots, 1);
            s.s_ccount.resize(sbase + m_slots, 0); s.s_parent.resize(sbase + m_slots, -1); s.s_xout.resize(sbase + m_slots, -1);
            s.s_child.resize((size_t)(sbase + m_slots) * 4, -1);
            for (int q = 0; q < m_slots; ++q) s.s_wsub[sbase + q] = G.wsub[q / WAVE];
            // cells are numbered in slot order (the hr_imd tape relies on it: a super-step's row is written cell next to cell)
            std::vector<int> by_slot(m_slots, -1);
            for (int i = 0; i < m; ++i) by_slot[slot_of[i]] = i;
            for (int q = 0; q < m_slots; ++q) {
                const int i = by_slot[q];
                if (i < 0) continue;
                const int node = q_node[i];
                if (node >= 0) { k_of_a[node] = knext++; s.s_cell[sbase + q] = k_of_a[node]; }
                else s.s_cell[sbase + q] = -1 - xslot_of[-1 - node];
                s.s_stage[sbase + q] = stage[i];
                s.s_sub[sbase + q] = sub[i];
                const int nk = kid_begin[i + 1] - kid_begin[i];
                s.s_ccount[sbase + q] = nk;
                unsigned short e[8];
                for (int j = 0; j < 8; ++j) e[j] = 0xffff;
                for (int j = 0; j < nk; ++j) { const int c = kids[kid_begin[i] + j]; e[j] = (unsigned short)(slot_of[c] | (merged[c] ? 0x8000 : 0)); }
                for (int j = 0; j < 4; ++j) s.s_child[(size_t)(sbase + q) * 4 + j] = (int)((unsigned)e[2 * j] | ((unsigned)e[2 * j + 1] << 16));
                s.s_parent[sbase + q] = q_par[i] < 0 ? -1 : (slot_of[q_par[i]] | (merged[i] ? 0x40000000 : 0));
                s.s_xout[sbase + q] = (node >= 0 && q_par[i] < 0) ? xslot_of[node] : -1;
                if (node >= 0 && q_par[i] < 0 && xslot_of[node] >= 0) s.x_prod_group[xslot_of[node]] = gi;
                if (node < 0) s.x_cons_group[xslot_of[-1 - node]] = gi;
            }
            s.g_slot_begin.push_back((int)s.s_cell.size());
        }
        s.round_group_begin.push_back((int)s.g_slot_begin.size() - 1);
    }
    s.ngroups = (int)s.g_slot_begin.size() - 1;
    s.nslots = (int)s.s_cell.size();
    if (knext != n) { s.error = "internal: cell numbering"; return -1; }

    s.cell_flat.resize(n);
    s.k_of_flat.assign(n2, -1);
    for (int a = 0; a < n; ++a) { s.cell_flat[k_of_a[a]] = flat_of_a[a]; s.k_of_flat[flat_of_a[a]] = k_of_a[a]; }
    s.gauge_k.resize(ng);
    for (int g = 0; g < ng; ++g) {
        const int row = gauge_pos[g], col = gauge_pos[g + ng];
        if (row < 0 || row >= nrow || col < 0 || col >= ncol || s.k_of_flat[row + (long)col * nrow] < 0) {
            s.error = "gauge outside the active domain"; return -1;
        }
        s.gauge_k[g] = s.k_of_flat[row + (long)col * nrow];
    }
    return 0;
}
