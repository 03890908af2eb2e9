// sx_plan.h -- host-side routing schedule: turns the D8 mesh (MeshDT flwdir / flwacc / active_cell,
// reference mwd_mesh.f90:45-72) into the tree partition the routing kernels run on.
//
// The reference visits cells in increasing flow-accumulation order inside every time step
// (md_forward_structure.f90:57-59, mesh%path from smash/mesh/meshing.py:216-224) because a cell reads
// the *current-step* discharge of its D8-upstream neighbours (md_routing_operator.f90:35-56).  On the
// GPU that dependency is honoured by cutting the river tree into subtrees of at most `group_size`
// members, each routed by ONE workgroup as a time-skewed wavefront through LDS, and by ordering the
// subtrees in "rounds": round r subtrees only receive inflow from subtrees of rounds < r, through
// per-inlet discharge series in HBM.  See DESIGN.md "Routing schedule".
#pragma once

#include <string>
#include <vector>

struct SxSchedule {
    int nrow = 0, ncol = 0;
    int n = 0;                       // active cells
    int group_size = 0;              // M: slots per workgroup
    std::vector<int> cell_flat;      // k -> row + col*nrow   (device cell order)
    std::vector<int> k_of_flat;      // flat -> k, -1 inactive
    // routing groups; groups of one round are contiguous
    int nrounds = 0, ngroups = 0, nslots = 0, nxslots = 0, max_stage = 0;
    std::vector<int> round_group_begin;  // nrounds + 1
    std::vector<int> g_slot_begin;       // ngroups + 1
    std::vector<int> g_dmax;             // ngroups: largest stage in the group
    // per slot (a slot = one thread of the routing workgroup: a real cell or an inlet pseudo-cell)
    std::vector<int> s_cell;     // k (>= 0) for a cell, -1 - x for the inlet fed by exchange series x
    // Stages and sub-levels (round 3).  The slots of a group are cut into COMPONENTS: connected pieces of the tree, at most `sublevels`
    // levels high, that live in ONE wavefront (64 consecutive slots).  All slots of a component share a stage; inside it a slot's
    // sub-level is one above its highest child of the same component.  A child is therefore either in its parent's component (same
    // stage, same wave, lower sub-level: its value passes through LDS inside the super-step, no workgroup barrier needed) or the
    // root of another component exactly one stage below.  A chain of L cells needs ~L / sublevels stages instead of L: the fill of a
    // routing launch -- (longest cell path) super-steps -- shrinks by that factor.  sublevels = 1 is the old schedule (every slot
    // its own component).
    std::vector<int> s_stage;
    std::vector<int> s_sub;      // sub-level inside the component (0 = no child in the same component)
    std::vector<int> s_wsub;     // number of sub-levels used in the slot's wavefront (the same for its 64 slots): the kernels' loop bound
    std::vector<int> s_child;    // 4 words per slot: 8 children in D8 order 1..8, 16 bits each: group-local slot index | 0x8000 if the child
                                 // is in the same component; unused entries 0xffff
    std::vector<int> s_ccount;
    std::vector<int> s_parent;   // group-local index of the parent (| 0x40000000 if it is in the same component), -1 for a subtree root
    std::vector<int> s_xout;     // subtree roots: exchange series id they publish, else -1
    std::vector<int> gauge_k;    // ng: device cell of every gauge
    // per exchange series: the group whose subtree root publishes it / the group holding its inlet slot
    // (-1: the other end lives in another tile).  Lets consecutive rounds run inside one launch.
    std::vector<int> x_prod_group, x_cons_group;
    // tile decomposition (multi-GPU): discharge series that cross the tile boundary, sorted by source cell
    std::vector<int> out_x, out_src, out_dst;   // series this tile publishes for a receiver in another tile (flat indices)
    std::vector<int> in_x, in_src, in_dst;      // series this tile needs from a cell in another tile
    std::string error;
};

// Returns 0 on success, negative on failure (s.error set).  All arrays column-major (row fastest).
// rect = {row0, row1, col0, col1} (half-open) restricts the schedule to the cells of one tile of the grid;
// cells of other tiles that drain into it become inlets fed by received series, cells draining out of it
// publish theirs.  rect == nullptr: the whole grid.
// own (nullable, (nrow,ncol), 1 = owned) replaces rect for arbitrary partitions (sub-catchments).
int sx_build_schedule(int nrow, int ncol, const int* flwdir, const int* active_cell, int ng, const int* gauge_pos,
                      int group_size, const int* rect, SxSchedule& s, const int* own = nullptr, int sublevels = 1);
