// K1g -- exact nearest-neighbour search through a uniform cell grid over the fixed cloud, with the box hierarchy (nn_tree.h) as
// the in-kernel fallback for the lanes the grid cannot serve cheaply.  The default search since round 2 (nn_grid.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "nn_tree.h"

namespace mislam {

constexpr int GRID_BLOCK = 64;             // moving points per workgroup: one wave = one row of ICP partial sums (icp_rows.hpp)
constexpr int GRID_MAX_DIM = 1024;         // cells per axis: keeps the rounding of a cell coordinate (2^-23 * 1024) far below the 1e-3 slack
constexpr float GRID_POINTS_PER_CELL = 1.25f;       // round 3's scan liked 1.5 (profiles/r03_scan_forms.log); with row_occ and the dealt rows: 1e6 points 0.0824 ms per search at 1.25 against 0.0845 at 1.5 (five interleaved runs each), 1e7 0.796 against 0.814, 1e5 the same; 1.0: as 1.25; 0.75 and 2: worse (profiles/r04_search_experiments.log)
#ifndef MISLAM_GRID_DU_MAX
#define MISLAM_GRID_DU_MAX 2.0f
#endif
constexpr float GRID_DU_MAX = MISLAM_GRID_DU_MAX;        // the grid answers queries whose neighbour lies within this many cells; the others walk the hierarchy
constexpr int GRID_STATS_ROWS = 1024;         // mi_profile_search_stats: the counters are kept in this many copies
constexpr int GRID_STATS_COLS = 32;           // counters per copy: [0, 8) mi_profile_search_stats, [8, 27) mi_profile_search_phases (nn_grid.hip)
#ifndef MISLAM_GRID_FAR_FACTOR
#define MISLAM_GRID_FAR_FACTOR 2.0f
#endif
constexpr float GRID_FAR_FACTOR = MISLAM_GRID_FAR_FACTOR;    // a starting candidate farther than this many times GRID_DU_MAX cells: straight to the hierarchy
constexpr int GRID_DEAL_ROWS_MIN_POINTS = 900000;   // from this many moving points on, a wave's leftover rows are dealt out one per lane (nn_grid.hip; measured: -6 % at 7e5, +2 % at 1e6, +5 % at 3e6)
// helper waves of the fused search (nn_grid.hip) by the number of moving points: up to GRID_HELPER_FULL_MAX_POINTS for every walk (mode 1),
// up to GRID_HELPER_MAX_POINTS only beside a scan (mode 2: chunks that walk at once keep to one wave), none beyond.  Measured search time,
// modes 0 / 1 / 2: 1e5 0.0370 / 0.0281 / 0.0301 ms, 3e5 0.0487 / 0.0440 / 0.0425, 4.5e5 0.0545 / 0.0581 / 0.0536, 7e5 0.0649 / 0.0828 / 0.0755
constexpr int GRID_HELPER_FULL_MAX_POINTS = 200000;
constexpr int GRID_HELPER_MAX_POINTS = 450000;
constexpr int GRID_CAND_BUDGET = 640;         // candidates a lane may test in the grid before it walks the hierarchy instead
constexpr int GRID_BATCH = 4;                 // cell rows whose offsets a lane requests together, then scans as one flat run of candidates
constexpr int GRID_PTS_PAD = 4;               // pts carries this many copies of its last entry: a lane fetches candidates four at a time
// cells (Chebyshev) the scan of a query can reach from the query's own (clamped) cell: its radius is capped a hair below
// GRID_DU_MAX cells (nn_grid.hip), so floor(u +- du) stays within ceil(GRID_DU_MAX) cells
constexpr int GRID_REACH_CELLS = (int)GRID_DU_MAX + ((float)(int)GRID_DU_MAX < GRID_DU_MAX ? 1 : 0);
#ifndef MISLAM_GRID_WALK_ONLY_MIN
#define MISLAM_GRID_WALK_ONLY_MIN 32
#endif
constexpr int GRID_WALK_ONLY_MIN = MISLAM_GRID_WALK_ONLY_MIN;
#ifndef MISLAM_GRID_COLD_PASSES
#define MISLAM_GRID_COLD_PASSES 6
#endif
constexpr int GRID_COLD_PASSES = MISLAM_GRID_COLD_PASSES;    // iterations of a registration whose starting candidates count as stale (nn_grid.hip: walk order, reach of the scan)   // lanes of a chunk beyond the grid's reach from which the chunk skips the scan next time (> 64: never)

struct NnGridView {
    const float4* pts;                     // the m fixed points sorted by cell (row-major: x fastest), w = GLOBAL index bits; + GRID_PTS_PAD
                                           // copies of the last one
    const unsigned int* cell_start;        // nx*ny*nz + 1 offsets into pts
    const unsigned int* slot_of;           // local index of a fixed point (global - index_base) -> its slot in pts
    const unsigned int* row_occ;           // per cell (cx, cy, cz): bit oz * 5 + oy set iff the cell ROW (cy + oy - 2, cz + oz - 2) holds a fixed point in
                                           // cells cx - 2 .. cx + 2 (GRID_REACH_CELLS).  0 tells a query in that cell that its scan cannot meet a single
                                           // candidate; a clear bit, that the scan need not look at that row (round 4)
    int index_base;                        // global index of this shard's point 0
    float ox, oy, oz;                      // lower corner of the bounding box
    float inv_h;                           // cells per unit length; cell coordinate of p on an axis: floor((p - o) * inv_h)
    float h_lo;                            // a hair less than 1 / inv_h: turns a gap in cells into a safe lower bound in length
    int nx, ny, nz;
};

// cell size and counts for a cloud of m points with bounding box bbox[6] (lo xyz, hi xyz); host-side, the same fp32 expression
// as the device's cell coordinate so that no point lands beyond the last cell
void grid_plan(const float bbox[6], int m, float points_per_cell, NnGridView* out);

struct GridBuildArgs {
    const float *x, *y, *z;                // the fixed-cloud shard, SoA
    int m;
    int index_base;                        // global index of point 0
    NnGridView view;                       // planned dims; pts / cell_start are written
    unsigned int* cell_fill;               // nx*ny*nz scratch words
    unsigned int* scan_tmp;                // scan scratch: (n_cells + 1) / 1024 + 2 words
    float4* pts_out;
    unsigned int* cell_start_out;
    unsigned int* slot_of_out;
    unsigned int* row_occ_out;             // nx*ny*nz words (NnGridView::row_occ)
    unsigned char* near_tmp;               // nx*ny*nz bytes of scratch
};
hipError_t grid_build(const GridBuildArgs& a, hipStream_t s);

// What one launch of the search works on.  Plain form: moving points from sx/sy/sz, keys in/out.  Fused ICP form (state != null):
// the whole O(N) part of an ICP iteration rides with the search -- see nn_grid.hip.
struct GridSearchArgs {
    // plain
    const float *sx, *sy, *sz;
    const int* done_flag;
    // both
    int n;
    unsigned long long* keys;
    unsigned long long* stats;             // null, or GRID_STATS_ROWS x 8 counters: candidates tested in the grid, cell rows scanned, lanes that went to
                                           // the hierarchy, lanes (mi_profile_search_stats)
    // fused ICP iteration
    IcpState* state;
    const float *bx, *by, *bz;             // the ORIGINAL moving cloud (Morton-sorted), SoA
    unsigned int* match_slot;              // per moving point: slot (NnGridView::pts) of its current match, ~0u = none.  The fixed
                                           // cloud's caller order is random, so gathering matches from it costs a cache line per
                                           // point; pts is sorted by cell and neighbours' matches share lines
    int shard_lo, shard_hi;
    int filter_pairs;
    float max_distance_squared;
    double* rows;                          // [ceil(n / GRID_BLOCK)][ICP_ROW] partial sums (icp_rows.hpp)
    const int* order;                      // work order (IcpSchedule): workgroup at position p takes chunk order[p]; null = identity
    unsigned char* far;                    // out: far[chunk] = this chunk's wave walked the hierarchy
    unsigned long long* far_lanes;         // in/out, may be null: per chunk, the lanes whose answer came from a walk (the helper wave's share next time)
    int split_walks;                       // fused iterations of small clouds: a helper wave per workgroup takes half of every immediate walk (nn_grid.hip)
    int deal_rows;                         // set by nn_grid_query: leftover rows dealt out one per lane (launches of more waves than the chip holds)
    int extend_reach, extend_reach_next;   // fused iterations: this search / the next one is among the first GRID_COLD_PASSES of its registration (the HOST's count of
                                           // enqueued iterations: the wider reach for queries outside the grid's extent is compiled in or out, nn_grid.hip grid_lane_cap2)
};
hipError_t nn_grid_query(const NnGridView& g, const NnTreeView& t, const GridSearchArgs& a, int fma, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
const char* nn_grid_kernel_name(bool fused);

}  // namespace mislam
