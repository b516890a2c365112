// K1g -- EXACT nearest-neighbour search through a uniform cell grid over the fixed cloud, the default search.
//
// Same contract as the every-pair K1 (nn_kernel.hip) and therefore as FindCorrespondences (cudacommon.cu:57-77) /
// common.cpp:446-462: idx[i] = argmin_j |after[j] - before[i]|^2, strict '<', lowest index on ties, the distance evaluated with
// the same fp32 operation sequence; results are IDENTICAL to the every-pair search bit for bit (the two rules of nn_walk.hpp).
//
// Why a grid.  Round 1's box-hierarchy walk spent 0.49 ms per search at N = M = 1e6 on ~325 L1 line accesses per query: every
// lane chases ~17 levels of dependent, divergent 16-byte loads before it sees its first candidate, although in ICP the answer is
// almost always within a fraction of a point spacing of the starting candidate (the previous match under the new transform).  A
// cell grid reaches the candidates in ONE dependent step: cell coordinate = floor((p - origin) / h) is arithmetic, the points of
// a cell row are contiguous in memory, and a row of cells [ix0, ix1] is one contiguous run pts[cell_start[r + ix0] ..
// cell_start[r + ix1 + 1]).  A query with search radius r visits the (2r/h + 1)^2 rows its sphere touches and tests
// ~(2r + h)^3 * density points: 2 offset loads + a short streaming loop per row, all lanes busy with their own candidates.
//
// Exactness.  Cell coordinates are compared in "cell units" u = (p - o) * inv_h, the SAME fp32 expression for fixed points (at
// build) and for queries, and fp32 rounding is monotonic: a point with p_x in [q_x - r, q_x + r] has u in [u(q_x - r), u(q_x + r)].
// Every rounding on the way is covered by explicit slack: a relative 1e-5 on radii and cell size (>> 2^-22), and 1e-3 cells on
// every cell-unit quantity (>> 2^-23 * GRID_MAX_DIM, the rounding of u itself).  A row is skipped only if its gap to the query,
// shrunk by that slack, squared and summed like a distance, is STRICTLY greater than the running best.  Visiting too many cells
// is harmless (lexicographic minimum), so every slack errs on that side.
//
// Fallback.  The grid is cheap while r is a few cells and cells hold a few points.  A lane without a starting candidate yet, far
// outside the fixed cloud, or in a crowded cell (clustered clouds, a far outlier stretching the bounding box) would test
// thousands of candidates: it stops instead (nothing within GRID_DU_MAX cells, or GRID_CAND_BUDGET candidates tested), and the
// lanes of a wave that stopped walk the box hierarchy together (nn_walk.hpp) from their current bests -- real candidates --
// inside the same launch.  Early ICP iterations are therefore mostly hierarchy walks, the rest mostly grid.
//
// Fused ICP iteration (GridSearchArgs::state != null).  The search needs the current position of every moving point and its
// previous match re-evaluated as starting candidate; the previous iteration's error needs exactly those two things, and the
// next solve needs the new matches' coordinates, which the search has just touched.  So ONE kernel does, per moving point:
// cur = R b + t (TransformCloud, cudacommon.cu:132-136; glm operation order) -> e = |a_old - cur|^2 -> error sums of the
// iteration just applied (GetMeanSquaredError, :138-148) -> search from (e, old match) -> key -> moments of the new pair
// (LeastSquaresSVD's sums, :168-201) -> one row of partial sums per workgroup (icp_rows.hpp).  The transformed cloud is never
// written: 20 bytes read + 8 written per moving point besides the search's own traffic.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>

#include "icp_rows.hpp"
#include "kernels.h"
#include "nn_grid.h"
#include "nn_walk.hpp"

namespace mislam {

// ---------------------------------------------------------------------------------------------------------------
// plan + build
// ---------------------------------------------------------------------------------------------------------------
// cell coordinate in cell units; the ONE expression used for fixed points, queries and the plan
__host__ __device__ __forceinline__ float cell_u(float p, float o, float inv_h) { return (p - o) * inv_h; }

void grid_plan(const float bbox[6], int m, float points_per_cell, NnGridView* out)
{
    double ext[3];
    bool active[3];
    for (int a = 0; a < 3; a++) {
        ext[a] = (double)bbox[3 + a] - (double)bbox[a];
        active[a] = ext[a] > 0.0 && std::isfinite(ext[a]);
    }
    const double cells = std::max(1.0, (double)m / (double)points_per_cell);
    double h = 1.0;
    for (int pass = 0; pass < 4; pass++) {
        int d = 0;
        double vol = 1.0;
        for (int a = 0; a < 3; a++)
            if (active[a]) { d++; vol *= ext[a]; }
        if (d == 0) { h = 1.0; break; }
        h = std::pow(vol / cells, 1.0 / d);
        bool changed = false;
        for (int a = 0; a < 3; a++)
            if (active[a] && ext[a] < h) { active[a] = false; changed = true; }   // a flat axis: one layer of cells, spend them elsewhere
        if (!changed) break;
    }
    double ext_max = 0.0;
    for (int a = 0; a < 3; a++)
        if (std::isfinite(ext[a])) ext_max = std::max(ext_max, ext[a]);
    if (ext_max > 0.0) h = std::max(h, ext_max / (GRID_MAX_DIM - 2));
    if (!(h > 0.0) || !std::isfinite(h)) h = 1.0;
    float inv_h = (float)(1.0 / h);
    int dims[3];
    for (int tries = 0; tries < 8; tries++) {
        bool ok = true;
        for (int a = 0; a < 3; a++) {
            const float umax = cell_u(bbox[3 + a], bbox[a], inv_h);          // the largest cell coordinate any point can get
            const int n = std::isfinite(umax) && umax > 0.f ? (int)floorf(umax) + 1 : 1;
            dims[a] = n;
            if (n > GRID_MAX_DIM) ok = false;
        }
        if (ok) break;
        inv_h *= 0.5f;
    }
    for (int a = 0; a < 3; a++) dims[a] = std::min(std::max(dims[a], 1), GRID_MAX_DIM);
    out->ox = bbox[0]; out->oy = bbox[1]; out->oz = bbox[2];
    out->inv_h = inv_h;
    out->h_lo = (1.0f / inv_h) * (1.0f - 1e-5f);
    out->nx = dims[0]; out->ny = dims[1]; out->nz = dims[2];
}

__device__ __forceinline__ int cell_index(float u, int n)
{
    return (int)fminf(fmaxf(floorf(u), 0.f), (float)(n - 1));
}

__device__ __forceinline__ unsigned int cell_of(const NnGridView& g, float x, float y, float z)
{
    const int ix = cell_index(cell_u(x, g.ox, g.inv_h), g.nx);
    const int iy = cell_index(cell_u(y, g.oy, g.inv_h), g.ny);
    const int iz = cell_index(cell_u(z, g.oz, g.inv_h), g.nz);
    return ((unsigned int)iz * (unsigned int)g.ny + (unsigned int)iy) * (unsigned int)g.nx + (unsigned int)ix;
}

__global__ __launch_bounds__(256) void grid_count_kernel(NnGridView g, const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ z, int m, unsigned int* __restrict__ counts)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    atomicAdd(&counts[cell_of(g, x[j], y[j], z[j])], 1u);
}

// Exclusive prefix sum of n words, 1024 per workgroup, in three small launches (tile sums -> scan of the tile sums -> tiles).
__global__ __launch_bounds__(256) void scan_tile_sums_kernel(const unsigned int* __restrict__ in, unsigned int n, unsigned int* __restrict__ sums)
{
    __shared__ unsigned int s[256];
    const unsigned int base = blockIdx.x * 1024u + threadIdx.x * 4u;
    unsigned int v = 0;
#pragma unroll
    for (unsigned int k = 0; k < 4; k++) v += base + k < n ? in[base + k] : 0u;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(1024) void scan_sums_kernel(unsigned int* __restrict__ sums, unsigned int count)
{
    __shared__ unsigned int s[1024];
    __shared__ unsigned int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (unsigned int base = 0; base < count; base += 1024u) {
        const unsigned int i = base + threadIdx.x;
        const unsigned int v = i < count ? sums[i] : 0u;
        s[threadIdx.x] = v;
        __syncthreads();
        for (unsigned int off = 1; off < 1024u; off <<= 1) {      // Hillis-Steele inclusive scan
            const unsigned int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0u;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < count) sums[i] = carry + s[threadIdx.x] - v;      // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void scan_tiles_kernel(const unsigned int* __restrict__ in, unsigned int n, const unsigned int* __restrict__ sums,
                                                         unsigned int* __restrict__ out, unsigned int* __restrict__ out_copy)
{
    __shared__ unsigned int s[256];
    const unsigned int base = blockIdx.x * 1024u + threadIdx.x * 4u;
    unsigned int v[4], tot = 0;
#pragma unroll
    for (unsigned int k = 0; k < 4; k++) { v[k] = base + k < n ? in[base + k] : 0u; tot += v[k]; }
    s[threadIdx.x] = tot;
    __syncthreads();
    for (unsigned int off = 1; off < 256u; off <<= 1) {
        const unsigned int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0u;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    unsigned int run = sums[blockIdx.x] + s[threadIdx.x] - tot;
#pragma unroll
    for (unsigned int k = 0; k < 4; k++) {
        if (base + k < n) {
            out[base + k] = run;
            if (out_copy != nullptr) out_copy[base + k] = run;
        }
        run += v[k];
    }
}

// Places point j in its cell's run.  The order inside a cell is whatever the atomics make it -- irrelevant to the result: the
// search takes a lexicographic (distance, GLOBAL index) minimum.
__global__ __launch_bounds__(256) void grid_scatter_kernel(NnGridView g, const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ z, int m, int index_base, unsigned int* __restrict__ fill,
                                                           float4* __restrict__ pts, unsigned int* __restrict__ slot_of)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const float px = x[j], py = y[j], pz = z[j];
    const unsigned int pos = atomicAdd(&fill[cell_of(g, px, py, pz)], 1u);
    pts[pos] = make_float4(px, py, pz, __int_as_float(j + index_base));
    slot_of[j] = pos;
}

// pts[m .. m + GRID_PTS_PAD) = copies of the last sorted point: the scan fetches four candidates at a time and may run up to three
// entries past a row's end -- every entry it can meet is a real point (testing one twice, or one of another cell, changes nothing)
__global__ void grid_pad_kernel(float4* __restrict__ pts, int m)
{
    if (threadIdx.x < GRID_PTS_PAD) pts[m + threadIdx.x] = pts[m - 1];
}

// row_occ (NnGridView): first, per cell, whether its run of x-neighbours -- one contiguous stretch of pts: two offsets tell -- holds a point;
// then, per cell, those flags of the 5 x 5 rows around it as one word
static_assert(GRID_REACH_CELLS == 2, "row_occ holds the 5 x 5 rows within two cells");
__global__ __launch_bounds__(256) void grid_near_x_kernel(NnGridView g, const unsigned int* __restrict__ cell_start, unsigned int n_cells,
                                                          unsigned char* __restrict__ out)
{
    const unsigned int c = blockIdx.x * 256u + threadIdx.x;
    if (c >= n_cells) return;
    const int x = (int)(c % (unsigned int)g.nx);
    const unsigned int row = c - (unsigned int)x;
    const int lo = max(x - GRID_REACH_CELLS, 0), hi = min(x + GRID_REACH_CELLS, g.nx - 1);
    out[c] = cell_start[row + hi + 1] > cell_start[row + lo] ? 1 : 0;
}

__global__ __launch_bounds__(256) void grid_row_occ_kernel(NnGridView g, const unsigned char* __restrict__ near_x, unsigned int n_cells,
                                                           unsigned int* __restrict__ out)
{
    const unsigned int c = blockIdx.x * 256u + threadIdx.x;
    if (c >= n_cells) return;
    const int x = (int)(c % (unsigned int)g.nx), y = (int)((c / (unsigned int)g.nx) % (unsigned int)g.ny), z = (int)(c / ((unsigned int)g.nx * (unsigned int)g.ny));
    unsigned int w = 0u;
#pragma unroll
    for (int oz = 0; oz < 5; oz++)
#pragma unroll
        for (int oy = 0; oy < 5; oy++) {
            const int iy = y + oy - 2, iz = z + oz - 2;
            if (iy >= 0 && iy < g.ny && iz >= 0 && iz < g.nz && near_x[((unsigned int)iz * (unsigned int)g.ny + (unsigned int)iy) * (unsigned int)g.nx + (unsigned int)x] != 0)
                w |= 1u << (oz * 5 + oy);
        }
    out[c] = w;
}

hipError_t grid_build(const GridBuildArgs& a, hipStream_t s)
{
    const NnGridView& g = a.view;
    const unsigned int n_cells = (unsigned int)g.nx * (unsigned int)g.ny * (unsigned int)g.nz;
    const unsigned int n_scan = n_cells + 1u;                    // the last entry becomes m
    hipError_t e = hipMemsetAsync(a.cell_fill, 0, sizeof(unsigned int) * (size_t)n_scan, s);
    if (e != hipSuccess) return e;
    const int pb = (a.m + 255) / 256;
    hipLaunchKernelGGL(grid_count_kernel, dim3(pb), dim3(256), 0, s, g, a.x, a.y, a.z, a.m, a.cell_fill);
    const unsigned int tiles = (n_scan + 1023u) / 1024u;
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(tiles), dim3(256), 0, s, a.cell_fill, n_scan, a.scan_tmp);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s, a.scan_tmp, tiles);
    // counts -> offsets, in place in cell_fill (the scatter's running cursors) and copied to cell_start
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(tiles), dim3(256), 0, s, a.cell_fill, n_scan, a.scan_tmp, a.cell_fill, a.cell_start_out);
    hipLaunchKernelGGL(grid_scatter_kernel, dim3(pb), dim3(256), 0, s, g, a.x, a.y, a.z, a.m, a.index_base, a.cell_fill, a.pts_out, a.slot_of_out);
    hipLaunchKernelGGL(grid_pad_kernel, dim3(1), dim3(64), 0, s, a.pts_out, a.m);
    const unsigned int cb = (n_cells + 255u) / 256u;
    hipLaunchKernelGGL(grid_near_x_kernel, dim3(cb), dim3(256), 0, s, g, a.cell_start_out, n_cells, a.near_tmp);
    hipLaunchKernelGGL(grid_row_occ_kernel, dim3(cb), dim3(256), 0, s, g, a.near_tmp, n_cells, a.row_occ_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// query
// ---------------------------------------------------------------------------------------------------------------
// gap, in cells, between cell coordinate u and the slab [i, i+1) of cells, shrunk by the slack: a lower bound
__device__ __forceinline__ float gap_cells(float u, int i)
{
    const float lo = (float)i;
    return fmaxf(fmaxf(lo - u, u - (lo + 1.f)) - 1e-3f, 0.f);
}

// ---- the grid part (round 3; round 2's lockstep slab loops: git show 67995b1:cuda-slam_amd/csrc/nn_grid.hip) --------------------
// What round 2's scan cost (profiles/r03_wave_timeline.log, r03_scan_forms.log): a launch lasts as long as its waves hold their slots, and a scan
// wave lived ~35 us through ~18 DEPENDENT memory round trips (every cell row: its two offsets, then its points, four at a time)
// while two thirds of its vector instructions were the bookkeeping of the lockstep slab loops.  Now:
//   * rows are named by a per-lane BIT MASK over the (2R+1)^2 rows around the query's own (R = GRID_REACH_CELLS): building it is
//     arithmetic (two ranges, one multiply), and the wave steps through set bits only -- a row no lane needs costs nothing;
//   * FIRST the 2 x 2 x 2 block of cells nearest to the query (its own cell and, on each axis, the neighbour on the side the query
//     leans to): with 1.25 - 1.5 points per cell the neighbour is in there nine times out of ten, so the radius is tight before anything
//     else is looked at -- whatever the starting candidate was worth; THEN the rows the shrunken radius still reaches, minus what
//     the block covered;
//   * GRID_BATCH rows at a time: their offsets are requested together (one round trip) and their points are ONE flat sequence of
//     runs, four candidates per trip -- and the trips are DEALT OUT over the wave (grid_deal_scan): ceil(total / 64) passes instead
//     of as many trips as the busiest lane has; the lockstep trip loop remains for batches where dealing does not pay or fit;
//   * a trip fetches pts[p .. p + 3] whatever is left of the run: what follows a run in pts are real points too (GRID_PTS_PAD copies
//     of the last one at the very end), and testing ANY real point is harmless for a lexicographic minimum;
//   * (distance, index) is ONE unsigned 64-bit key -- non-negative floats order like their bit patterns -- so "closer, or as
//     close with a lower index" is one compare.
constexpr int GRID_ROWS_R = GRID_REACH_CELLS;
constexpr int GRID_ROWS_W = 2 * GRID_ROWS_R + 1;
static_assert(GRID_ROWS_W * GRID_ROWS_W <= 32, "row masks are 32 bits wide: GRID_DU_MAX <= 2");
constexpr unsigned int grid_rows_rep()
{
    unsigned int r = 0;
    for (int k = 0; k < GRID_ROWS_W; k++) r |= 1u << (GRID_ROWS_W * k);
    return r;
}
// bit (oz * W + oy) for every row offset (oy, oz) in [y0, y1] x [z0, z1] (0 <= lo <= hi < W): an outer product, i.e. one multiply
__device__ __forceinline__ unsigned int grid_rows_mask(int y0, int y1, int z0, int z1)
{
    y0 = min(max(y0, 0), GRID_ROWS_W - 1); y1 = min(max(y1, -1), GRID_ROWS_W - 1);     // (shift counts stay in range whatever comes in;
    z0 = min(max(z0, 0), GRID_ROWS_W - 1); z1 = min(max(z1, -1), GRID_ROWS_W - 1);     //  hi < lo gives an empty mask)
    const unsigned int ym = ((1u << (y1 + 1)) - 1u) & ~((1u << y0) - 1u);
    const unsigned int zm = grid_rows_rep() & ((1u << (GRID_ROWS_W * (z1 + 1))) - 1u) & ~((1u << (GRID_ROWS_W * z0)) - 1u);
    return ym * zm;
}

struct GridLane {
    float q[3];
    float gx2;                           // squared gap (a lower bound) between the query and the grid's extent along x: every row's test starts from it
    float u1, u2;                        // the query's cell coordinates on y and z
    int cy, cz;                          // its own row (clamped into the grid)
    unsigned long long kbest;            // fp32 bits of the best distance << 32 | its global index
    unsigned int bslot;                  // slot in pts of the best candidate met by the scan
    int budget;
    bool alive;
    unsigned int n_rows;                 // (STATS)
    // (STATS) wave-uniform trip counts of the scan's loops, what tools/isa_budget.py multiplies the kernel's static instruction counts by:
    // [0] block batches run, [1] of them dealt, [2] deal passes, [3] iterations of the deal's write loop, [4] lockstep trips;
    // [5] leftover rounds (rows dealt out), [6] of them with their trips dealt, [7] deal passes, [8] write-loop iterations, [9] lockstep trips,
    // [10] leftover batches of the four-rows-per-lane form, [11] waves that had leftover rows
    unsigned int ph[12];
#ifdef MISLAM_DEV_WAVE_TIMELINE
    unsigned long long t_in, t_block;    // developer build: entry of the scan, end of its first batch
    unsigned int trips_block, trips_rest, batches_rest;
#endif
};

#ifndef MISLAM_GRID_TRIP
#define MISLAM_GRID_TRIP 4
#endif
constexpr unsigned int GRID_TRIP = MISLAM_GRID_TRIP;          // candidates per trip (a power of two, <= GRID_PTS_PAD)
static_assert(GRID_TRIP >= 1 && GRID_TRIP <= GRID_PTS_PAD && (GRID_TRIP & (GRID_TRIP - 1)) == 0, "trip width");
__device__ __forceinline__ unsigned int grid_trip_round(int c) { return (unsigned int)(c + (int)GRID_TRIP - 1) & ~(GRID_TRIP - 1u); }

// A batch's trips dealt out over the wave.  The lockstep trip loop below lasts as long as the lane with the MOST trips has any (8.9
// for the block's four rows, where the mean lane has 4.6): every one of them is four gathers and ~65 vector instructions for the whole
// wave.  Here each lane first says how many trips it has (n < 32), the wave lays them end to end (prefix sum by ballots), every lane
// writes the slots of its trips into LDS, and then lane L takes trip L, L + 64, ... -- whoever it belongs to: it reads the owner's
// query and running best from LDS, tests the four candidates, and folds the result back with an LDS minimum.  ceil(total / 64) passes
// instead of max(n) trips; which lane tests a candidate changes nothing (a lexicographic minimum over the same candidates).
#ifndef MISLAM_GRID_DEAL_MAX
#define MISLAM_GRID_DEAL_MAX 384
#endif
#ifndef MISLAM_GRID_DEAL_GAIN
#define MISLAM_GRID_DEAL_GAIN 1          // deal when the passes (plus this) are fewer than the longest lane's trips
#endif
constexpr unsigned int GRID_DEAL_MAX = MISLAM_GRID_DEAL_MAX;
// The wave's LDS (one wave per workgroup; 4.9 KB: 32 workgroups fit a CU's 160 KB)
struct GridWaveLds {
    unsigned int deal_p[GRID_DEAL_MAX];              // trip -> slot of its first candidate
    unsigned char deal_owner[GRID_DEAL_MAX];         // trip -> the lane it belongs to
    float4 deal_q[64];                               // per OWNER: its query (w: the radius its leftover rows are tested with)
    unsigned long long deal_key[64];                 // per OWNER: running minimum of the keys found for it
    unsigned int deal_slot[64];                      // per OWNER: where that minimum sits in pts
    // the leftover rows dealt out (grid_rows_dealt): one record per lane that has rows, packed in lane order
    unsigned int rec_first[64];                      // rank -> number of rows before this lane's | lane << 16
    unsigned int rec_rows[64];                       // rank -> its row mask
    unsigned int own_x[64];                          // per OWNER: cell range x0 | x1 << 16
    unsigned int own_c[64];                          // per OWNER: its own row cy | cz << 16
    unsigned int own_cand[64];                       // per OWNER: candidates its rows hold (the budget)
};
__device__ __forceinline__ GridWaveLds& grid_wave_lds()
{
    __shared__ GridWaveLds lds;
    return lds;
}
// The dealing arrays belong to ONE wave (a workgroup's second wave, where there is one, never touches them): what orders their traffic
// is the wave's own program order -- LDS operations of a wave complete in issue order -- so the compiler must not move them across
// these points, and no workgroup barrier is needed (a barrier here would also tie the scan to the helper wave's walk).
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// Returns false (nothing done) if dealing does not pay or does not fit; true: kbest / bslot hold the lane's results.
// `owner_lane`: the lane the trips are tested FOR (the lane itself, or -- leftover rows dealt out -- the lane whose row it scans; then
// `own_setup` is false: the owners' queries and keys are in LDS already).
template <bool FMA, bool COUNT = false>
__device__ __forceinline__ bool grid_deal_scan(const float4* __restrict__ pts, const float q[3], unsigned int n_trips, unsigned int e1, unsigned int e2,
                                               unsigned int e3, unsigned int b0, unsigned int b1, unsigned int b2, unsigned int b3,
                                               unsigned long long& kbest, unsigned int& bslot, unsigned int& dev_passes,
                                               unsigned int owner_lane, bool own_setup, unsigned int* count = nullptr)
{
    static_assert(GRID_TRIP == 4, "a dealt trip is four candidates");
    GridWaveLds& L = grid_wave_lds();
    unsigned int* deal_p = L.deal_p; unsigned char* deal_owner = L.deal_owner; float4* deal_q = L.deal_q;
    unsigned long long* deal_key = L.deal_key; unsigned int* deal_slot = L.deal_slot;
    const int lane = (int)threadIdx.x & 63;
    if (__builtin_amdgcn_ballot_w64(n_trips >= 32u) != 0ull) return false;
    // counts -> exclusive prefix, total and maximum, bit by bit: ballots, mbcnt and scalar arithmetic, no exchange
    const unsigned long long B0 = __builtin_amdgcn_ballot_w64((n_trips & 1u) != 0u), B1 = __builtin_amdgcn_ballot_w64((n_trips & 2u) != 0u),
                             B2 = __builtin_amdgcn_ballot_w64((n_trips & 4u) != 0u), B3 = __builtin_amdgcn_ballot_w64((n_trips & 8u) != 0u),
                             B4 = __builtin_amdgcn_ballot_w64((n_trips & 16u) != 0u);
    const unsigned int total = (unsigned int)__builtin_popcountll(B0) + 2u * (unsigned int)__builtin_popcountll(B1) + 4u * (unsigned int)__builtin_popcountll(B2) +
                               8u * (unsigned int)__builtin_popcountll(B3) + 16u * (unsigned int)__builtin_popcountll(B4);
    unsigned int longest = 0u;
    {
        unsigned long long cand = ~0ull;                        // lanes that still can hold the maximum
        const unsigned long long B[5] = {B0, B1, B2, B3, B4};
#pragma unroll
        for (int b = 4; b >= 0; b--) {
            const unsigned long long m = B[b] & cand;
            if (m != 0ull) { longest |= 1u << b; cand = m; }
        }
    }
    const unsigned int passes = (total + 63u) >> 6;
    if (total > GRID_DEAL_MAX || passes + MISLAM_GRID_DEAL_GAIN >= longest) return false;
    auto below = [](unsigned long long m) { return (unsigned int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u)); };
    const unsigned int first = below(B0) + 2u * below(B1) + 4u * below(B2) + 8u * below(B3) + 16u * below(B4);
    if (own_setup) {
        deal_q[lane] = make_float4(q[0], q[1], q[2], 0.f);
        deal_key[lane] = kbest;
    }
    const unsigned char mine = (unsigned char)owner_lane;
    if (COUNT) { count[0] += 1u; count[1] += (total + 63u) >> 6; count[2] += longest; }
    for (unsigned int k = 0; k < longest; k++) {                // (wave-uniform bound: the lane with the most trips)
        if (k < n_trips) {
            const unsigned int t = k * GRID_TRIP;
            deal_p[first + k] = (t >= e2 ? (t >= e3 ? b3 : b2) : (t >= e1 ? b1 : b0)) + t;
            deal_owner[first + k] = mine;
        }
    }
    wave_lds_sync();                                            // (the arrays are one wave's own: orders its LDS traffic)
    for (unsigned int base = 0; base < total; base += 64u) {
        const unsigned int idx = base + (unsigned int)lane;
        const bool have = idx < total;
        const unsigned int p = have ? deal_p[idx] : 0u;
        const unsigned int owner = have ? (unsigned int)deal_owner[idx] : (unsigned int)lane;
        const float4 oq = deal_q[owner];
        unsigned long long kb = deal_key[owner];
        const unsigned long long kb0 = kb;
        unsigned int slot = 0u;
        const float4* __restrict__ pp = pts + p;
        const float4 cs[4] = {pp[0], pp[1], pp[2], pp[3]};
#pragma unroll
        for (int j4 = 0; j4 < 4; j4++) {
            const float d = sq3<FMA>(cs[j4].x - oq.x, cs[j4].y - oq.y, cs[j4].z - oq.z);
            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(cs[j4].w);
            const bool better = key < kb;
            kb = better ? key : kb;
            slot = better ? p + (unsigned int)j4 : slot;
        }
        const bool won = have && kb < kb0;
        if (won) atomicMin(&deal_key[owner], kb);
        // LDS operations of one wave complete in program order: every lane now reads the minimum of the pass, and whoever holds it says where
        if (won && deal_key[owner] == kb) deal_slot[owner] = slot;
#ifdef MISLAM_DEV_WAVE_TIMELINE
        dev_passes += 1u;
#endif
    }
    wave_lds_sync();
    const unsigned long long kfin = deal_key[lane];
    if (kfin < kbest) { kbest = kfin; bslot = deal_slot[lane]; }
    return true;
}

// One batch: takes up to GRID_BATCH rows off `mask`, scans cells [x0, x1] of each that lies within r2.  BLOCK: the rows are the (up to)
// 2 x 2 rows [blk.y0, blk.y1] x [blk.z0, blk.z1] of the nearest block instead -- named directly, their two gaps per axis computed once.
struct GridBlockRows { int y0, y1, z0, z1; };
// `reach` (BLOCK only): row_occ of the query's cell, a word that is still on its way from memory when the block's offsets are
// requested -- it only decides whether their runs count, so the two round trips overlap.
template <bool FMA, bool STATS, bool BLOCK>
__device__ __forceinline__ void grid_batch(const NnGridView& g, GridLane& s, unsigned int& mask, int x0, int x1, float r2, const GridBlockRows& blk,
                                           unsigned int reach_word = 1u)
{
    const float4* __restrict__ pts = g.pts;
    const unsigned int* __restrict__ cell_start = g.cell_start;
    unsigned int S[GRID_BATCH];
    int C[GRID_BATCH];
    float gy2b[2] = {0.f, 0.f}, gz2b[2] = {0.f, 0.f};
    if (BLOCK) {
        const float a = gap_cells(s.u1, blk.y0) * g.h_lo, b = gap_cells(s.u1, blk.y1) * g.h_lo, c = gap_cells(s.u2, blk.z0) * g.h_lo, d = gap_cells(s.u2, blk.z1) * g.h_lo;
        gy2b[0] = a * a; gy2b[1] = b * b; gz2b[0] = c * c; gz2b[1] = d * d;
    }
#pragma unroll
    for (int j = 0; j < GRID_BATCH; j++) {
        bool have;
        int iy, iz;
        float g2;                                              // the row's squared gap to the query, a lower bound (summed like a distance)
        if (BLOCK) {
            static_assert(GRID_BATCH == 4, "the block's 2 x 2 rows are one batch");
            have = ((j & 1) == 0 || blk.y1 > blk.y0) && ((j & 2) == 0 || blk.z1 > blk.z0);
            iy = (j & 1) ? blk.y1 : blk.y0; iz = (j & 2) ? blk.z1 : blk.z0;
            g2 = (gy2b[j & 1] + gz2b[(j >> 1) & 1]) + s.gx2;
        } else {
            have = mask != 0u;
            const int b = have ? __builtin_ctz(mask) : 0;
            mask &= mask - 1u;
            const int oz = (b * ((256 + GRID_ROWS_W - 1) / GRID_ROWS_W)) >> 8, oy = b - GRID_ROWS_W * oz;   // b / W, b % W (b < 64)
            iy = s.cy + oy - GRID_ROWS_R; iz = s.cz + oz - GRID_ROWS_R;
            const float gy = gap_cells(s.u1, iy) * g.h_lo, gz = gap_cells(s.u2, iz) * g.h_lo;
            g2 = (gy * gy + gz * gz) + s.gx2;
        }
        // a row is skipped only if it is strictly farther than the search radius: then it cannot win or tie
        const bool ok = have && s.alive && x1 >= x0 && g2 <= r2;
        const unsigned int rb = ((unsigned int)iz * (unsigned int)g.ny + (unsigned int)iy) * (unsigned int)g.nx;
        // (not ok: both offsets of entry 0, an empty run.  Two independent 4-byte loads per row, eight per batch, all in flight together:
        // one 16-byte load per row of the block was tried -- the compiler narrows it to the words the cell range selects, behind a
        // branch with a wait inside, four memory round trips in a row instead of one)
        S[j] = cell_start[ok ? rb + (unsigned int)x0 : 0u];
        C[j] = (int)(cell_start[ok ? rb + (unsigned int)x1 + 1u : 0u] - S[j]);
        if (STATS) s.n_rows += ok ? 1u : 0u;
    }
    if (BLOCK) {
        // the byte is looked at only now, behind the offsets (loads return in order: it has arrived when they have) -- the compiler would
        // otherwise test it the moment it is loaded and stall the wave a whole round trip before the offsets are even requested
        asm("" : "+v"(reach_word) : "v"(C[0]), "v"(C[1]), "v"(C[2]), "v"(C[3]));      // (not volatile: see far_class)
        const bool reach = reach_word != 0u;
#pragma unroll
        for (int j = 0; j < GRID_BATCH; j++) C[j] = reach ? C[j] : 0;
        s.alive = s.alive && reach;
    }
#pragma unroll
    for (int j = 0; j < GRID_BATCH; j++) {
        if (C[j] > s.budget) s.alive = false;                  // crowded: give up, the hierarchy takes over
        C[j] = s.alive ? C[j] : 0;
        s.budget -= C[j];
    }
    // The batch's runs as ONE flat sequence of candidates, four per trip.  What a launch waits for is the wave's INSTRUCTION count, not
    // its gathers (profiles/r03_scan_forms.log: with every gather of the scan replaced by register moves the launch takes the same
    // time), so the loop is built for few instructions: each run is rounded up to whole trips (what follows a run in pts are real
    // points, and testing any real point is harmless), the runs are laid end to end on ONE position counter t, and the run a trip
    // belongs to falls out of three compares -- no queue to shift, no lane masked off: a lane that has run out keeps re-testing
    // its last four candidates.
    static_assert(GRID_BATCH == 4, "four runs laid end to end");
    const unsigned int e1 = grid_trip_round(C[0]), e2 = e1 + grid_trip_round(C[1]), e3 = e2 + grid_trip_round(C[2]), e4 = e3 + grid_trip_round(C[3]);
    const unsigned int b0 = S[0], b1 = S[1] - e1, b2 = S[2] - e2, b3 = S[3] - e3;      // slot of position t inside run r: b_r + t
    const unsigned int t_last = e4 >= GRID_TRIP ? e4 - GRID_TRIP : 0u;
    unsigned long long kbest = s.kbest;
    unsigned int bslot = s.bslot;
#ifndef MISLAM_GRID_NO_DEAL
    {
        unsigned int dev_passes = 0u;
        if (STATS) s.ph[BLOCK ? 0 : 10] += 1u;
        const bool dealt = grid_deal_scan<FMA, STATS>(pts, s.q, e4 / GRID_TRIP, e1, e2, e3, b0, b1, b2, b3, kbest, bslot, dev_passes, (unsigned int)threadIdx.x & 63u, true,
                                                      STATS ? &s.ph[BLOCK ? 1 : 6] : nullptr);
#ifdef MISLAM_DEV_WAVE_TIMELINE
        if (BLOCK) s.trips_block += dev_passes; else s.trips_rest += dev_passes;
#endif
        if (dealt) { s.kbest = kbest; s.bslot = bslot; return; }
    }
#endif
    for (unsigned int t = 0; __builtin_amdgcn_ballot_w64(t < e4) != 0ull; t += GRID_TRIP) {
        if (STATS) s.ph[BLOCK ? 4 : 9] += 1u;
#ifdef MISLAM_DEV_WAVE_TIMELINE
        if (BLOCK) s.trips_block += 1; else s.trips_rest += 1;
#endif
        const unsigned int tt = min(t, t_last);
        const unsigned int p = (tt >= e2 ? (tt >= e3 ? b3 : b2) : (tt >= e1 ? b1 : b0)) + tt;
        const float4* __restrict__ pp = pts + p;               // one address, three immediate offsets
        float4 cs[GRID_TRIP];
#pragma unroll
        for (unsigned int j4 = 0; j4 < GRID_TRIP; j4++) cs[j4] = pp[j4];
#pragma unroll
        for (unsigned int j4 = 0; j4 < GRID_TRIP; j4++) {
            const float d = sq3<FMA>(cs[j4].x - s.q[0], cs[j4].y - s.q[1], cs[j4].z - s.q[2]);
            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(cs[j4].w);
            const bool better = key < kbest;                   // d >= +0: closer, or as close with a lower index
            kbest = better ? key : kbest;
            bslot = better ? p + (unsigned int)j4 : bslot;
        }
    }
    s.kbest = kbest;
    s.bslot = bslot;
}

// The leftover rows of a wave DEALT OUT, one row per lane.  Taken four per lane and batch (grid_batch) a wave needed as many batches
// as its busiest lane has rows / 4 -- 2.3 on average (profiles/r04_wave_timeline.log), ~240 vector instructions each for everyone,
// while most lanes have no row left at all.  Here every lane that has rows writes ONE record (its mask, and how many rows the lanes
// before it hold: a prefix sum by ballots), and lane L takes row L of the wave -- it finds the record by a binary search over the
// prefix sums, the row as the n-th set bit of the record's mask, and tests it for its OWNER: the owner's query, radius, cell range
// and own row come from LDS, the result goes back through an LDS minimum, exactly as for dealt trips.  Every row is tested with the
// radius the owner had when the rows were dealt (grid_batch re-reads it between batches): a superset, never a miss.
// Fewer instructions, more dependent steps (two barriers, a binary search over LDS): +2 % iterations/s at 1e6 points, +5 % at 3e6, where
// the launch is several times what the chip holds and instruction issue is what it waits for -- and -6 % at 7e5, -12 % at 1e5 and 1e4,
// where (nearly) every wave is resident from the start and the launch lasts as long as ONE wave's chain of dependent steps
// (profiles/r04_search_experiments.log).  Hence GridSearchArgs::deal_rows: by size, GRID_DEAL_ROWS_MIN_POINTS.
#ifndef MISLAM_GRID_ROWS_DEALT
#define MISLAM_GRID_ROWS_DEALT 1
#endif
__device__ __forceinline__ int nth_set_bit(unsigned int m, unsigned int n)      // position of set bit number n (from 0) of m; n < popcount(m)
{
    int pos = 0;
    unsigned int c = (unsigned int)__builtin_popcount(m & 0xffffu);
    if (n >= c) { n -= c; pos = 16; m >>= 16; }
    c = (unsigned int)__builtin_popcount(m & 0xffu);
    if (n >= c) { n -= c; pos += 8; m >>= 8; }
    c = (unsigned int)__builtin_popcount(m & 0xfu);
    if (n >= c) { n -= c; pos += 4; m >>= 4; }
    c = (unsigned int)__builtin_popcount(m & 3u);
    if (n >= c) { n -= c; pos += 2; m >>= 2; }
    if (n >= (m & 1u)) pos += 1;
    return pos;
}
template <bool FMA, bool STATS>
__device__ __forceinline__ void grid_rows_dealt(const NnGridView& g, GridLane& s, unsigned int mask, int x0, int x1, float r2)
{
    GridWaveLds& L = grid_wave_lds();
    const float4* __restrict__ pts = g.pts;
    const unsigned int* __restrict__ cell_start = g.cell_start;
    const unsigned int lane = (unsigned int)threadIdx.x & 63u;
    if (!s.alive || x1 < x0) mask = 0u;
    const unsigned int k = (unsigned int)__builtin_popcount(mask);              // <= 25
    const unsigned long long B0 = __builtin_amdgcn_ballot_w64((k & 1u) != 0u), B1 = __builtin_amdgcn_ballot_w64((k & 2u) != 0u),
                             B2 = __builtin_amdgcn_ballot_w64((k & 4u) != 0u), B3 = __builtin_amdgcn_ballot_w64((k & 8u) != 0u),
                             B4 = __builtin_amdgcn_ballot_w64((k & 16u) != 0u);
    const unsigned int total = (unsigned int)__builtin_popcountll(B0) + 2u * (unsigned int)__builtin_popcountll(B1) + 4u * (unsigned int)__builtin_popcountll(B2) +
                               8u * (unsigned int)__builtin_popcountll(B3) + 16u * (unsigned int)__builtin_popcountll(B4);
    auto below = [](unsigned long long m) { return (unsigned int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u)); };
    const unsigned int first = below(B0) + 2u * below(B1) + 4u * below(B2) + 8u * below(B3) + 16u * below(B4);
    const unsigned long long producers = B0 | B1 | B2 | B3 | B4;
    const unsigned int n_rec = (unsigned int)__builtin_popcountll(producers), rank = below(producers);
    L.deal_q[lane] = make_float4(s.q[0], s.q[1], s.q[2], r2 - s.gx2);       // (.w: what a row's y-z gap is tested against -- the x face gap taken off)
    L.deal_key[lane] = s.kbest;
    L.own_x[lane] = (unsigned int)x0 | ((unsigned int)x1 << 16);
    L.own_c[lane] = (unsigned int)s.cy | ((unsigned int)s.cz << 16);
    L.own_cand[lane] = 0u;
    if (mask != 0u) { L.rec_first[rank] = first | (lane << 16); L.rec_rows[rank] = mask; }
    wave_lds_sync();                                            // (the arrays are one wave's own: orders its LDS traffic)
    if (STATS) s.ph[11] += 1u;
    for (unsigned int base = 0; base < total; base += 64u) {    // (wave-uniform; one round unless the wave has more than 64 rows left)
        if (STATS) s.ph[5] += 1u;
        const unsigned int c = base + lane;
        const bool have = c < total;
        unsigned int lo = 0u;                                   // the last record that starts at or before row c
#pragma unroll
        for (unsigned int w = 32u; w != 0u; w >>= 1) {
            const unsigned int j = lo + w;
            const unsigned int f = L.rec_first[j & 63u] & 0xffffu;
            lo = (j < n_rec && f <= c) ? j : lo;
        }
        const unsigned int rec = L.rec_first[lo], rows = have ? L.rec_rows[lo] : 1u;
        const unsigned int owner = have ? rec >> 16 : lane;
        const int b = nth_set_bit(rows, have ? c - (rec & 0xffffu) : 0u);
        const int oz = (b * ((256 + GRID_ROWS_W - 1) / GRID_ROWS_W)) >> 8, oy = b - GRID_ROWS_W * oz;   // b / W, b % W
        const float4 oq = L.deal_q[owner];
        const unsigned int ox = L.own_x[owner], oc = L.own_c[owner];
        const int iy = (int)(oc & 0xffffu) + oy - GRID_ROWS_R, iz = (int)(oc >> 16) + oz - GRID_ROWS_R;
        const float gy = gap_cells(cell_u(oq.y, g.oy, g.inv_h), iy) * g.h_lo, gz = gap_cells(cell_u(oq.z, g.oz, g.inv_h), iz) * g.h_lo;
        // a row is skipped only if it is strictly farther than the search radius: then it cannot win or tie
        const bool ok = have && gy * gy + gz * gz <= oq.w;
        const unsigned int rb = ((unsigned int)iz * (unsigned int)g.ny + (unsigned int)iy) * (unsigned int)g.nx;
        const unsigned int S = cell_start[ok ? rb + (ox & 0xffffu) : 0u];
        int C = (int)(cell_start[ok ? rb + (ox >> 16) + 1u : 0u] - S);
        if (STATS) s.n_rows += ok ? 1u : 0u;
        // crowded (one row beyond a lane's whole budget): not scanned, and the owner gives up -- the hierarchy takes over
        const bool crowded = C > GRID_CAND_BUDGET;
        if (ok) atomicAdd(&L.own_cand[owner], crowded ? (unsigned int)(2 * GRID_CAND_BUDGET) : (unsigned int)C);
        C = crowded ? 0 : C;
        const unsigned int e = grid_trip_round(C);
        const unsigned int t_last = e >= GRID_TRIP ? e - GRID_TRIP : 0u;
#ifdef MISLAM_DEV_WAVE_TIMELINE
        s.batches_rest += 1;
#endif
#ifndef MISLAM_GRID_NO_DEAL
        {
            // the trips of the dealt rows, dealt in turn (3.6 lockstep trips per round without, 1.6 passes + trips with: the rows' runs
            // differ in length); kbest / bslot: the lane's OWN -- the deal ends with every lane collecting what was found for it
            unsigned int dev_passes = 0u;
            const bool dealt = grid_deal_scan<FMA, STATS>(pts, s.q, e / GRID_TRIP, e, e, e, S, S, S, S, s.kbest, s.bslot, dev_passes, owner, false, STATS ? &s.ph[6] : nullptr);
#ifdef MISLAM_DEV_WAVE_TIMELINE
            s.trips_rest += dev_passes;
#endif
            if (dealt) continue;
        }
#endif
        unsigned long long kb = L.deal_key[owner];
        const unsigned long long kb0 = kb;
        unsigned int slot = 0u;
        for (unsigned int t = 0; __builtin_amdgcn_ballot_w64(t < e) != 0ull; t += GRID_TRIP) {
            if (STATS) s.ph[9] += 1u;
#ifdef MISLAM_DEV_WAVE_TIMELINE
            s.trips_rest += 1;
#endif
            const unsigned int p = S + min(t, t_last);
            const float4* __restrict__ pp = pts + p;
            float4 cs[GRID_TRIP];
#pragma unroll
            for (unsigned int j4 = 0; j4 < GRID_TRIP; j4++) cs[j4] = pp[j4];
#pragma unroll
            for (unsigned int j4 = 0; j4 < GRID_TRIP; j4++) {
                const float d = sq3<FMA>(cs[j4].x - oq.x, cs[j4].y - oq.y, cs[j4].z - oq.z);
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(cs[j4].w);
                const bool better = key < kb;
                kb = better ? key : kb;
                slot = better ? p + j4 : slot;
            }
        }
        const bool won = C > 0 && kb < kb0;
        if (won) atomicMin(&L.deal_key[owner], kb);
        // LDS operations of one wave complete in program order: every lane now reads the minimum, and whoever holds it says where
        if (won && L.deal_key[owner] == kb) L.deal_slot[owner] = slot;
        wave_lds_sync();
    }
    wave_lds_sync();
    const unsigned long long kfin = L.deal_key[lane];
    if (kfin < s.kbest) { s.kbest = kfin; s.bslot = L.deal_slot[lane]; }
    const int cand = (int)L.own_cand[lane];
    if (cand > s.budget) s.alive = false;
    s.budget -= min(cand, s.budget);
    wave_lds_sync();                                            // (the records are read: the next use of the arrays may write)
}

// How far the scan of a query reaches (round 5).  The scan looks at the cells within GRID_REACH_CELLS of the query's own (clamped) cell on every
// axis -- the width of the row masks and of row_occ.  For a query INSIDE the grid's extent that covers every point within two cells: cap = 2 h,
// rounds 2-4's rule for everybody.  A query OUTSIDE the extent by g cells along an axis (early and middle iterations: a shell of the moving
// cloud still hangs out of the fixed cloud's box, one chunk in five at iteration 20) has all its candidates on ONE side: a point within r of
// it differs from it by at least g on that axis, hence by at most sqrt(r^2 - g^2) on the two others -- the lens the sphere cuts out of the
// cloud's face is two cells wide for r^2 <= 4 + g^2.  In general a point within r lies, on axis a, within sqrt(r^2 - sum_{b != a} g_b^2) of the
// query, of which g_a is spent before the extent begins: the cells it can be in stay within two of the clamped own cell iff
//   r^2 <= (g_a + 2)^2 + sum_{b != a} g_b^2   on every axis a,
// in cell units, every g shrunk by the rounding slack.  Such lanes used to walk the hierarchy whatever the shape of their lens (cap 2 h):
// 37 % of all wave-time at iteration 20 for 16 % of the lanes (profiles/r04_wave_timeline.log).  Beyond 16 cells outside the relative
// slacks of the cell arithmetic outgrow the absolute ones: cap 0, those lanes walk as before.
#ifndef MISLAM_GRID_EXTENT_REACH
#define MISLAM_GRID_EXTENT_REACH 1       // 0 (tools/build_variant.sh): rounds 2-4's rule -- a cap of two cells for every query
#endif
// `extend` (wave-uniform): WHEN the wider reach is used.  Measured (profiles/r05_search_experiments.log, the bench's two cubes with their roles
// swapped, so that the moving cloud's shell hangs out of the fixed cloud's extent): it pays while the starting candidates are poor -- a search
// without any (the plain mi_nn_search), the first iterations of a registration (0.289 -> 0.237 ms per search in iterations 0-4, walking lanes
// 77 -> 72 %) -- and LOSES once they are good (iterations 20-24: walking lanes 16.6 -> 5.7 %, search 0.084 -> 0.101 ms): a lens across the row
// direction is one boundary cell per row, ten rows per lane where an inside lane scans 3.6, and the walk it replaces is then a short verification.
// So: plain searches and the cold iterations of a fused registration only.
__device__ __forceinline__ float grid_lane_cap2(const NnGridView& g, float u0, float u1, float u2, float (&gs)[3], bool extend)
{
#if MISLAM_GRID_EXTENT_REACH
    gs[0] = extend ? fmaxf(fmaxf(-u0, u0 - (float)g.nx) - 1e-3f, 0.f) : 0.f;
    gs[1] = extend ? fmaxf(fmaxf(-u1, u1 - (float)g.ny) - 1e-3f, 0.f) : 0.f;
    gs[2] = extend ? fmaxf(fmaxf(-u2, u2 - (float)g.nz) - 1e-3f, 0.f) : 0.f;
#else
    (void)u0; (void)u1; (void)u2; (void)extend;
    gs[0] = gs[1] = gs[2] = 0.f;
#endif
    const float e = GRID_DU_MAX - 2e-3f;
    const float s0 = gs[0] * gs[0], s1 = gs[1] * gs[1], s2 = gs[2] * gs[2];
    const float c0 = (gs[0] + e) * (gs[0] + e) + (s1 + s2), c1 = (gs[1] + e) * (gs[1] + e) + (s0 + s2), c2 = (gs[2] + e) * (gs[2] + e) + (s0 + s1);
    const float far_out = fmaxf(gs[0], fmaxf(gs[1], gs[2]));
    const float cells2 = far_out <= 16.f ? fminf(c0, fminf(c1, c2)) : 0.f;
    return cells2 * (g.h_lo * g.h_lo) * (1.f - 1e-5f);
}
__device__ __forceinline__ float grid_lane_cap2(const NnGridView& g, const float q[3], bool extend)
{
    float gs[3];
    return grid_lane_cap2(g, cell_u(q[0], g.ox, g.inv_h), cell_u(q[1], g.oy, g.inv_h), cell_u(q[2], g.oz, g.inv_h), gs, extend);
}
// cells (as a float count) a radius^2 of r2 spans on an axis once the squared face gaps of the two OTHER axes are taken off it; the slack of
// the cell arithmetic added (raw v_sqrt_f32, 1 ulp: covered by it)
__device__ __forceinline__ float grid_du(const NnGridView& g, float r2, float other_gaps2)
{
    return __builtin_amdgcn_sqrtf(fmaxf(r2 - other_gaps2, 0.f) * 1.000001f) * g.inv_h * 1.00001f + 1e-3f;
}

// Grid part of one lane's search.  Returns true if the lane gave up (it must then walk the hierarchy from (best, bidx)).
//
// The lane tests the points of every cell row within r2 = min(best, cap2) of the query; every point with d <= the FINAL r2 is
// met, because r2 only shrinks and every row test and cell range uses an r2 that is at least the final one.  So if the final best
// is within cap2 the answer is exact; otherwise (nothing near: the query lies outside the fixed cloud, or has no starting
// candidate and sits in an empty region) the lane gives up.  cap2 is the square of (a hair less than) GRID_DU_MAX cells.
// `reach_word` = row_occ of the query's cell: 0 says no cell the scan could visit holds a point, a clear bit that its row holds none.  `lane_on`: the lane has a point.
template <bool FMA, bool STATS>
__device__ __forceinline__ bool grid_search(const NnGridView& g, const float q[3], bool lane_on, unsigned int reach_word, float& best, unsigned int& bidx,
                                            unsigned int& bslot, unsigned int& n_cand, unsigned int& n_rows, bool deal_rows, bool extend_reach, unsigned int (&phases)[12]
#ifdef MISLAM_DEV_WAVE_TIMELINE
                                            , unsigned long long (&dev_tl)[3]
#endif
                                            )
{
    const float u0 = cell_u(q[0], g.ox, g.inv_h), u1 = cell_u(q[1], g.oy, g.inv_h), u2 = cell_u(q[2], g.oz, g.inv_h);
    float gs[3];
    float cap2 = grid_lane_cap2(g, u0, u1, u2, gs, extend_reach);            // every point within sqrt(cap2) of the query lies within GRID_REACH_CELLS cells of its clamped own cell
    // (without the extended reach the cap is the same for every lane -- a product of wave-uniform floats, which only the vector unit can form: said
    // so, it waits for the end of the scan in a scalar register instead of a vector one the 72-register budget does not have)
    // (as inline assembly: the builtin is moved up to h_lo^2 and the rest of the product goes back into a vector register)
    if (!extend_reach) {
        float cap2_uniform;
        asm("v_readfirstlane_b32 %0, %1" : "=s"(cap2_uniform) : "v"(cap2));
        cap2 = cap2_uniform;
    }
    // squared face gaps in length units, lower bounds (h_lo): what a point within r has already spent on an axis before the extent begins
    const float G0 = (gs[0] * g.h_lo) * (gs[0] * g.h_lo), G1 = (gs[1] * g.h_lo) * (gs[1] * g.h_lo), G2 = (gs[2] * g.h_lo) * (gs[2] * g.h_lo);
    GridLane s;
    s.gx2 = G0;
    s.q[0] = q[0]; s.q[1] = q[1]; s.q[2] = q[2];
    s.u1 = u1; s.u2 = u2;
    s.kbest = ((unsigned long long)__float_as_uint(best) << 32) | bidx;
    s.bslot = bslot;
    s.budget = GRID_CAND_BUDGET;
    s.n_rows = 0u;
    if (STATS) for (int k = 0; k < 12; k++) s.ph[k] = 0u;
    const int cx = cell_index(u0, g.nx);
    s.cy = cell_index(u1, g.ny);
    s.cz = cell_index(u2, g.nz);
    // The cells to visit for a radius: every point within sqrt(r2) of the query on an axis has its cell coordinate within du of
    // the query's (monotonic rounding + the slack), hence its cell index in [floor(u - du), floor(u + du)], clamped into the grid
    // like the points' own cell indices.  raw v_sqrt_f32 (1 ulp): covered by the slack.  du < GRID_DU_MAX, so every row lies
    // within GRID_ROWS_R rows of the query's own: the masks' width.
    float r2 = fminf(best, cap2);
    float dux = grid_du(g, r2, G1 + G2), duy = grid_du(g, r2, G0 + G2), duz = grid_du(g, r2, G0 + G1);
    float fx0 = floorf(u0 - dux), fx1 = floorf(u0 + dux), fy0 = floorf(u1 - duy), fy1 = floorf(u1 + duy), fz0 = floorf(u2 - duz), fz1 = floorf(u2 + duz);
    // (alive: geometry only so far; the reach byte joins in inside the first batch)
    s.alive = lane_on && fx1 >= 0.f && fx0 <= (float)(g.nx - 1) && fy1 >= 0.f && fy0 <= (float)(g.ny - 1) && fz1 >= 0.f && fz0 <= (float)(g.nz - 1);
    // ---- the nearest 2 x 2 x 2 block (what of it the starting radius reaches)
    const int sx = u0 - (float)cx >= 0.5f ? 1 : -1, sy = u1 - (float)s.cy >= 0.5f ? 1 : -1, sz = u2 - (float)s.cz >= 0.5f ? 1 : -1;
    const int xa0 = max(max(min(cx, cx + sx), 0), (int)fmaxf(fx0, 0.f)), xa1 = min(min(max(cx, cx + sx), g.nx - 1), (int)fminf(fx1, (float)(g.nx - 1)));
    const int ya0 = max(max(min(s.cy, s.cy + sy), 0), (int)fmaxf(fy0, 0.f)), ya1 = min(min(max(s.cy, s.cy + sy), g.ny - 1), (int)fminf(fy1, (float)(g.ny - 1)));
    const int za0 = max(max(min(s.cz, s.cz + sz), 0), (int)fmaxf(fz0, 0.f)), za1 = min(min(max(s.cz, s.cz + sz), g.nz - 1), (int)fminf(fz1, (float)(g.nz - 1)));
    // (alive: the query's own clamped cell lies inside the clamped ranges, so each of the three is non-empty)
    const unsigned int block = s.alive ? grid_rows_mask(ya0 - s.cy + GRID_ROWS_R, ya1 - s.cy + GRID_ROWS_R, za0 - s.cz + GRID_ROWS_R, za1 - s.cz + GRID_ROWS_R) : 0u;
    unsigned int mask = block;
#ifdef MISLAM_DEV_WAVE_TIMELINE
    s.t_in = wall_clock64(); s.trips_block = s.trips_rest = s.batches_rest = 0;
#endif
    const GridBlockRows blk{ya0, ya1, za0, za1};
    if (__builtin_amdgcn_ballot_w64(mask != 0u) != 0ull) grid_batch<FMA, STATS, true>(g, s, mask, xa0, xa1, r2, blk, reach_word);
    s.alive = s.alive && reach_word != 0u;
#ifdef MISLAM_DEV_WAVE_TIMELINE
    s.t_block = wall_clock64();
#endif
    // ---- what the radius as it stands now still reaches, minus the rows of the block if their cells were all covered
    r2 = fminf(__uint_as_float((unsigned int)(s.kbest >> 32)), cap2);
    dux = grid_du(g, r2, G1 + G2); duy = grid_du(g, r2, G0 + G2); duz = grid_du(g, r2, G0 + G1);
    fx0 = floorf(u0 - dux); fx1 = floorf(u0 + dux); fy0 = floorf(u1 - duy); fy1 = floorf(u1 + duy); fz0 = floorf(u2 - duz); fz1 = floorf(u2 + duz);
    {
        const int x0 = (int)fmaxf(fx0, 0.f), x1 = (int)fminf(fx1, (float)(g.nx - 1));
        const int y0 = (int)fmaxf(fy0, 0.f), y1 = (int)fminf(fy1, (float)(g.ny - 1));
        const int z0 = (int)fmaxf(fz0, 0.f), z1 = (int)fminf(fz1, (float)(g.nz - 1));
        const bool covered = x0 >= xa0 && x1 <= xa1;           // the block's rows need no second look
        // (& reach_word: rows that hold no point within the scan's cells are never looked at -- row_occ; round 3 found them empty one batch of four
        // at a time, 2.3 batches per scan wave where one lane in five has any rows left)
        mask = s.alive ? grid_rows_mask(y0 - s.cy + GRID_ROWS_R, y1 - s.cy + GRID_ROWS_R, z0 - s.cz + GRID_ROWS_R, z1 - s.cz + GRID_ROWS_R) & ~(covered ? block : 0u) & reach_word : 0u;
    }
    // ---- the leftover rows -- about one lane in five has any (its neighbour lies beyond the block, or the radius still pokes out of it),
    // one to four each, more at the cloud's edge -- four per lane and round, their trips dealt out over the wave like the block's.  (Round 3
    // first dealt the ROWS out, one per lane, before there was a way to deal trips: git show eefc579:cuda-slam_amd/csrc/nn_grid.hip; with the
    // trips dealt the two are as fast, at 1e5, 1e6 and 1e7 points, and this is the shorter code.)
#if MISLAM_GRID_ROWS_DEALT
    if (deal_rows && __builtin_amdgcn_ballot_w64(mask != 0u) != 0ull) {
        const int x0 = (int)fmaxf(fx0, 0.f), x1 = (int)fminf(fx1, (float)(g.nx - 1));
        grid_rows_dealt<FMA, STATS>(g, s, mask, x0, x1, r2);
        mask = 0u;
    }
#endif
    while (__builtin_amdgcn_ballot_w64(mask != 0u) != 0ull) {
        // the cells of a row from the radius as it stands now: one range for the whole batch (a superset of what each row's own
        // gap would leave of it)
        r2 = fminf(__uint_as_float((unsigned int)(s.kbest >> 32)), cap2);
        dux = grid_du(g, r2, G1 + G2);
        const float flo = floorf(u0 - dux), fhi = floorf(u0 + dux);
#ifdef MISLAM_DEV_WAVE_TIMELINE
        s.batches_rest += 1;
#endif
        grid_batch<FMA, STATS, false>(g, s, mask, (int)fmaxf(flo, 0.f), (int)fminf(fhi, (float)(g.nx - 1)), r2, blk);
    }
    best = __uint_as_float((unsigned int)(s.kbest >> 32));
    bidx = (unsigned int)s.kbest;
    bslot = s.bslot;
    if (STATS) {
        n_cand += (unsigned int)(GRID_CAND_BUDGET - s.budget); n_rows += s.n_rows;
        for (int k = 0; k < 12; k++) phases[k] = s.ph[k];
    }
#ifdef MISLAM_DEV_WAVE_TIMELINE
    dev_tl[0] = s.t_in; dev_tl[1] = s.t_block; dev_tl[2] = s.trips_block | ((unsigned long long)s.trips_rest << 16) | ((unsigned long long)s.batches_rest << 32);
#endif
    return !s.alive || !(best <= cap2);
}

// One wave per workgroup, one lane per moving point: a wave's 64 Morton neighbours share cells and cache lines, and nothing has
// to be exchanged between waves -- the lanes that give up walk the hierarchy TOGETHER right where they are (tree_walk_wave takes
// any subset of a wave), the others wait masked off.
// waves per SIMD the register allocation must leave room for: a launch lasts as long as its waves hold their slots (the walking
// waves hold a third of them for most of it), so slots count for more than registers here
#ifndef MISLAM_GRID_XCD_RUN
#define MISLAM_GRID_XCD_RUN (TREE_XCD_CHUNKS * (256 / GRID_BLOCK))
#endif
#ifndef MISLAM_GRID_MIN_WAVES
#define MISLAM_GRID_MIN_WAVES 7
#endif
#define MI_GRID_OCC __attribute__((amdgpu_waves_per_eu(MISLAM_GRID_MIN_WAVES, 8)))
// WAVES == 2 (fused iterations of SMALL clouds, nn_grid_query): a HELPER wave per workgroup.  A launch of a small cloud fits the chip
// several times over and lasts as long as its longest wave -- at 1e5 points a chunk that scans and THEN walks for the lanes the grid cannot
// serve: 11 us + 14 us on average, 38 at most, where a scan-only wave takes 12 (profiles/r04_wave_timeline_1e5.log).  The helper takes the
// walks: in a chunk that scans, the lanes that ended beyond the grid's reach last time (far_lanes; they will again) walk THERE, at once,
// while the first wave scans for the others -- max(scan, walk) instead of scan + walk; in a chunk that walks at once (class 2) the two waves
// take half the lanes each.  The helper hands its answers over through LDS and returns; a chunk with nothing to help with loses its helper
// at once.  Everything after the walk is the first wave's, unchanged -- an exact walk and an exact scan find the same neighbour, so the
// keys and the rows are the same to the last bit (tests/test_gpu_icp.py).  Search at 1e4 / 1e5 / 3e5 points: 0.0245 -> 0.0178 / 0.0370 ->
// 0.0281 / 0.0487 -> 0.0425 ms; beyond 4.5e5 points the helpers cost more slots than they save time (nn_grid.h: modes by size).
// EXTEND (fused iterations; a plain search always has it): the scan's wider reach for queries outside the grid's extent -- compiled OUT of the warm
// iterations' kernel, where it loses (grid_lane_cap2) and its three face gaps would cost the 72-register budget more spills.
template <bool FMA, bool FUSED, bool STATS, int WAVES = 1, bool EXTEND = true>
__global__ __launch_bounds__(GRID_BLOCK * WAVES) MI_GRID_OCC void nn_grid_kernel(NnGridView g, NnTreeView t, GridSearchArgs a)
{
    static_assert(GRID_BLOCK == 64 && ICP_ROW_POINTS == 64, "one wave = one workgroup = one row of partial sums");
    static_assert(WAVES == 1 || (WAVES == 2 && FUSED && !STATS), "the helper wave exists for fused iterations only");
    if (FUSED) {
        if (as_constant(&a.state->done)[0] != 0) return;
    } else if (a.done_flag != nullptr && *a.done_flag != 0) return;
    const int tid = (int)threadIdx.x & 63;
    const int helper = WAVES == 2 ? (int)threadIdx.x >> 6 : 0;         // 1: the second wave of the workgroup
#ifdef MISLAM_DEV_WAVE_TIMELINE        // developer build: per wave { start, end of scan, end } in 100 MHz ticks + walk steps (tools/wave_timeline.py)
    const unsigned long long tl_start = wall_clock64();
    unsigned long long tl_scan = 0, dev_tl[3] = {0, 0, 0}, tl_p1 = 0, tl_p2 = 0, tl_p3 = 0;
#define MI_TL_STAMP(var, dep) do { asm volatile("" :: "v"(dep)); var = wall_clock64(); } while (0)
#else
#define MI_TL_STAMP(var, dep) do { } while (0)
#endif
    unsigned int chunk = xcd_chunk(blockIdx.x, gridDim.x, MISLAM_GRID_XCD_RUN);
    if (FUSED) chunk = (unsigned int)as_constant(a.order)[chunk];          // walking chunks first (IcpSchedule): speed only (read-only for the launch: a scalar load, nn_walk.hpp as_constant)
    if (WAVES == 2 && helper != 0) {
        // a helper with nothing to help with leaves before it has loaded anything else (three chunks in four, late in a registration)
        const unsigned int fc = a.far[chunk];
        const unsigned long long fl = a.far_lanes[chunk];
        if (fc < 2u ? fl == 0ull : a.split_walks == 2) return;   // (split_walks == 2: chunks that walk at once are left to their first wave)
    }
    const int i = (int)(chunk * GRID_BLOCK) + tid;
    const bool valid = i < a.n;
    MI_TL_STAMP(tl_p1, i);
    // the chunk's class from the last iteration (requested here, used below: its round trip overlaps the moving point's)
    // (through a per-lane address as far as the compiler can tell: it would move a wave-uniform byte into a scalar register, and wait
    // for it, before the next branch)
    unsigned int far_class = 0u;
    if (FUSED) {
        unsigned int lane_zero = 0u;
        asm("" : "+v"(lane_zero));                           // (not volatile: that would cost every later load its scalar form)
        far_class = a.far[chunk + lane_zero];
    }

    float q[3] = {0.f, 0.f, 0.f};
    float best = __builtin_inff();
    unsigned int bidx = 0u, bslot = ~0u;
    // the previous pair's squared error and whether it counts, carried to the epilogue as ONE float and a wave mask (two scalar registers) -- as two
    // doubles they took four vector registers across the whole search, which the 72-register budget paid for in scratch (16 bytes per lane)
    float e_prev = 0.f;
    unsigned long long kept_prev = 0ull;
    if (FUSED) {
        float R[9], tr[3];
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = as_constant(a.state->R)[k];
#pragma unroll
        for (int k = 0; k < 3; k++) tr[k] = as_constant(a.state->t)[k];
        bool kept_lane = false;
        if (valid) {
            const float x = a.bx[i], y = a.by[i], z = a.bz[i];
            // TransformPoint: (rotationMatrix * point) + translationVector  (common.cpp:45-49), glm operation order
            q[0] = ((R[0] * x + R[3] * y) + R[6] * z) + tr[0];
            q[1] = ((R[1] * x + R[4] * y) + R[7] * z) + tr[1];
            q[2] = ((R[2] * x + R[5] * y) + R[8] * z) + tr[2];
            MI_TL_STAMP(tl_p2, q[0] + q[1] + q[2]);
            const unsigned long long key = a.keys[i];
            const float d2 = __uint_as_float((unsigned int)(key >> 32));
            bslot = a.match_slot[i];
            if (bslot != ~0u) {                                 // there is a previous match
                const float4 p = g.pts[bslot];
                const int gidx = __float_as_int(p.w);
                const float dx = p.x - q[0], dy = p.y - q[1], dz = p.z - q[2];
                const float e = (dx * dx + dy * dy) + dz * dz;   // diff.LengthSquared(), common.cpp:264-265
                const bool kept = a.filter_pairs ? (d2 < a.max_distance_squared) : true;
                e_prev = kept ? e : 0.f;
                kept_lane = kept;
                // the old match under the new transform is a real candidate, evaluated with the search's own arithmetic
                best = FMA ? __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) : e;
                bidx = (unsigned int)gidx;
            }
        }
        kept_prev = __builtin_amdgcn_ballot_w64(kept_lane);
    } else if (valid) {
        q[0] = a.sx[i]; q[1] = a.sy[i]; q[2] = a.sz[i];
        unpack_start(a.keys[i], best, bidx);
    }

    MI_TL_STAMP(tl_p3, best);
    // row_occ of the query's (clamped) cell: 0 = the scan could not meet a single point; a clear bit = nothing in that row (requested here, ahead of the scan)
    // (every lane asks -- a lane without a point has q = 0, some cell of the grid: no branch, so nothing waits for the byte here)
    const unsigned int near_word = g.row_occ[cell_of(g, q[0], q[1], q[2])];
    bool hard = false;
    unsigned int n_cand = 0u, n_rows = 0u, n_nodes = 0u, n_leaves = 0u;
    unsigned int phases[12] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};      // (STATS: grid_search's loop trip counts, GridLane::ph)
    bool scanned = false;
    // A chunk most of whose lanes ended beyond the grid's reach last time (they will again: the flags move slowly) skips the scan:
    // all its lanes walk, each from its own starting candidate -- the walk is exact by itself, the few lanes the scan would have
    // served add little to the union the wave visits anyway, and the wave's critical path loses the scan (speed only).
    const bool walk_only = FUSED && (unsigned int)__builtin_amdgcn_readfirstlane((int)far_class) >= 2u;
    // the lanes of this chunk whose answer came from a walk last time (they will walk again: the flags move slowly).  With a helper wave
    // they walk THERE, at once, while the first wave scans for the others -- a chunk's chain is max(scan, walk) instead of scan + walk
    unsigned long long predicted = 0ull;
    if (WAVES == 2 && !walk_only) {
        const unsigned long long w = a.far_lanes[chunk];
        predicted = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(w >> 32)) << 32) |
                    (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)w);
    }
    const bool mine_to_walk = WAVES == 2 && ((predicted >> tid) & 1ull) != 0ull;   // (of a scanning chunk: this lane is the helper's)
    if (WAVES == 2 && helper != 0 && !walk_only && predicted == 0ull) return;   // (nothing to help with: the first wave scans alone, as ever)
    // the first iterations of a registration move the cloud by many cells: the starting candidates (previous matches) are STALE, and a walk
    // that enters the children in index order meets the true neighbourhood late; there the nearest child goes first (a lane vote per step) -- and
    // the scan reaches farther for queries outside the grid's extent (grid_lane_cap2), as it does for a plain search, which has no candidates at all
    const bool cold = FUSED && as_constant(&a.state->passes)[0] < GRID_COLD_PASSES;
    constexpr bool extend_reach = !FUSED || EXTEND;
    const bool halves = WAVES == 2 && a.split_walks != 2;                  // a chunk that walks at once: half the lanes per wave
    if (walk_only) hard = valid && (!halves || (tid >> 5) == helper);
    else if (WAVES == 2 && helper != 0) hard = valid && mine_to_walk;
#ifdef MISLAM_DEV_WAVE_TIMELINE
    else hard = grid_search<FMA, STATS>(g, q, valid && !mine_to_walk, near_word, best, bidx, bslot, n_cand, n_rows, a.deal_rows != 0, extend_reach, phases, dev_tl) && valid && !mine_to_walk;
#else
    else hard = grid_search<FMA, STATS>(g, q, valid && !mine_to_walk, near_word, best, bidx, bslot, n_cand, n_rows, a.deal_rows != 0, extend_reach, phases) && valid && !mine_to_walk;   // (all lanes: the loops run in step)
#endif
    if (STATS) scanned = !walk_only && !(WAVES == 2 && helper != 0);
    bool walked = __builtin_amdgcn_ballot_w64(hard) != 0ull;
#ifdef MISLAM_DEV_WAVE_TIMELINE
    tl_scan = wall_clock64();
#endif
    unsigned int walk_counts[5] = {0u, 0u, 0u, 0u, 0u};
    if (hard) tree_walk_wide<FMA, STATS>(t, q, best, bidx, n_nodes, n_leaves, cold, STATS ? walk_counts : nullptr);
    if (WAVES == 2 && (walk_only ? halves : predicted != 0ull)) {       // (workgroup-uniform: both waves are here)
        __shared__ float x_best[64];
        __shared__ unsigned int x_bidx[64];
        const bool helpers = walk_only ? tid >= 32 : mine_to_walk;      // the lanes whose walk the helper wave took
        if (helper != 0 && helpers) { x_best[tid] = best; x_bidx[tid] = bidx; }
        __syncthreads();
        if (helper != 0) return;
        if (helpers) { best = x_best[tid]; bidx = x_bidx[tid]; hard = valid; }   // (their answers come from a walk)
        walked = __builtin_amdgcn_ballot_w64(hard) != 0ull;
    }
    // the point's index again, from the chunk number (a scalar) as far as the compiler can tell a different one: kept from the prologue it
    // would sit in two vector registers through the whole search, which is short of them
    unsigned int chunk_again = chunk;
    asm("" : "+s"(chunk_again));
    const int io = (int)(chunk_again * GRID_BLOCK) + tid;
    if (valid && best < __builtin_inff()) a.keys[io] = ((unsigned long long)__float_as_uint(best) << 32) | bidx;
    // measurement hook (mi_profile_search_stats).  Kept BEHIND the walk: a global atomic ahead of it would stop the compiler from
    // using scalar loads for the hierarchy (it can no longer prove those arrays unwritten)
    if (STATS) {
        // spread over GRID_STATS_ROWS rows of 8 counters (64 bytes apart): atomics on one line serialise at ~10 ns each
        unsigned long long* srow = a.stats + (size_t)(blockIdx.x % GRID_STATS_ROWS) * GRID_STATS_COLS;
        unsigned int c0 = valid ? n_cand : 0u, c1 = valid ? n_rows : 0u;
        unsigned int v0 = hard ? n_nodes : 0u, v1 = hard ? n_leaves : 0u;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            c0 += __shfl_xor(c0, m, 64); c1 += __shfl_xor(c1, m, 64);
            v0 = max(v0, (unsigned int)__shfl_xor(v0, m, 64)); v1 = max(v1, (unsigned int)__shfl_xor(v1, m, 64));
        }
        const unsigned long long nh = __builtin_popcountll(__builtin_amdgcn_ballot_w64(hard)), nv = __builtin_popcountll(__builtin_amdgcn_ballot_w64(valid));
#ifdef MISLAM_DEV_WAVE_TIMELINE
        if (tid == 0) {
            unsigned long long* tl = a.stats + (size_t)GRID_STATS_ROWS * GRID_STATS_COLS + (size_t)blockIdx.x * 16;
            tl[8] = tl_p1; tl[9] = tl_p2; tl[10] = tl_p3;
            tl[0] = tl_start; tl[1] = tl_scan; tl[2] = wall_clock64();
            tl[3] = (unsigned long long)(v0 + v1) | (nh << 16) | ((unsigned long long)(walk_only ? 1 : 0) << 32) | ((unsigned long long)chunk << 40);
            tl[4] = dev_tl[0]; tl[5] = dev_tl[1]; tl[6] = dev_tl[2];
            tl[7] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);   // HW_ID, XCC_ID
        }
#endif
        if (tid == 0) {                                         // one set of atomics per wave
            atomicAdd(&srow[0], (unsigned long long)c0);
            atomicAdd(&srow[1], (unsigned long long)c1);
            atomicAdd(&srow[2], nh);
            atomicAdd(&srow[3], nv);
            if (walked) {                                       // hierarchy nodes and leaves visited, waves walking
                atomicAdd(&srow[4], (unsigned long long)v0);
                atomicAdd(&srow[5], (unsigned long long)v1);
                atomicAdd(&srow[6], 1ull);
                atomicMax(&srow[7], (unsigned long long)(v0 + v1));     // the longest walk, in steps
            }
            // the loops' trip counts (wave-uniform: lane 0's), mi_profile_search_phases: [8] waves, [9] waves that scanned, [10] walk-only waves, [11 ..] GridLane::ph
            atomicAdd(&srow[8], 1ull);
            if (scanned) atomicAdd(&srow[9], 1ull);
            if (walk_only) atomicAdd(&srow[10], 1ull);
            if (scanned)
                for (int k = 0; k < 12; k++)
                    if (phases[k] != 0u) atomicAdd(&srow[11 + k], (unsigned long long)phases[k]);
        }
        {
            // the walk's other loops (wave-uniform trip counts; the walking lanes agree, the others hold 0): a maximum over the wave, then one atomic each
            unsigned int w0 = hard ? walk_counts[0] : 0u, w1 = hard ? walk_counts[1] : 0u, w2 = hard ? walk_counts[2] : 0u, w3 = hard ? walk_counts[3] : 0u, w4 = hard ? walk_counts[4] : 0u;
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) {
                w0 = max(w0, (unsigned int)__shfl_xor(w0, m, 64)); w1 = max(w1, (unsigned int)__shfl_xor(w1, m, 64));
                w2 = max(w2, (unsigned int)__shfl_xor(w2, m, 64)); w3 = max(w3, (unsigned int)__shfl_xor(w3, m, 64));
                w4 = max(w4, (unsigned int)__shfl_xor(w4, m, 64));
            }
            if (tid == 0 && walked) {
                atomicAdd(&srow[23], (unsigned long long)w0);
                atomicAdd(&srow[24], (unsigned long long)w1);
                atomicAdd(&srow[25], (unsigned long long)w2);
                atomicAdd(&srow[26], (unsigned long long)w3);
                atomicAdd(&srow[27], (unsigned long long)w4);
            }
        }
    }
    if (FUSED) {
        bool use_pair = false;
        float4 pm = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid && best < __builtin_inff()) {
            const int gidx = (int)bidx;
            const bool mine = gidx >= a.shard_lo && gidx < a.shard_hi;
            const bool kept = a.filter_pairs ? (best < a.max_distance_squared) : true;
            if (hard) bslot = g.slot_of[gidx - g.index_base];   // a walk's winner: the hierarchy keeps its own order
            a.match_slot[io] = bslot;
            if (mine && kept) {
                pm = g.pts[bslot];
                use_pair = true;
            }
        } else if (valid) a.match_slot[io] = ~0u;
        double* row = a.rows + (size_t)chunk * ICP_ROW;
        row_store_pair_moments(use_pair, q[0], q[1], q[2], pm.x, pm.y, pm.z, row);      // (four fp64 matrix-pipe products: icp_rows.hpp)
        row_store_error(e_prev, ((kept_prev >> tid) & 1ull) != 0ull ? 1.f : 0.f, row);      // (fp32 in: the conversions happen inside, next to the products)
        {
            // next iteration's class of this chunk: 0 = no lane walked, 1 = some did, 2 = most lanes lie beyond the grid's reach
            // (this lane's reach under the rule the NEXT search will apply: the class is a prediction for it)
            // (the cell size through a scalar register the compiler cannot tell from the one the search used: h_lo^2 -- a wave-uniform product with
            // no scalar instruction to compute it -- otherwise sits in a vector register from the scan to here, across the whole search)
            NnGridView ge = g;
            asm("" : "+s"(ge.h_lo));
            // (the next search's rule: with the extended reach a function of the lane's position outside the extent -- some forty instructions --, without it
            // ONE number for the whole wave; which of the two is a launch argument, so the warm kernel branches around the arithmetic)
            float cap2;
            if (!FUSED || a.extend_reach_next != 0) cap2 = grid_lane_cap2(ge, q, true);
            else { float gs0[3]; cap2 = grid_lane_cap2(ge, 0.f, 0.f, 0.f, gs0, false); }
            const int beyond = (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(valid && !(best <= cap2)));
            if (tid == 0) a.far[chunk] = beyond >= GRID_WALK_ONLY_MIN ? 2 : (walked ? 1 : 0);
            // (the helper wave's share next time: the lanes the scan cannot serve -- NOT "the lanes that walked", which would keep every lane
            // that ever walked walking for good)
            const unsigned long long out_of_reach = __builtin_amdgcn_ballot_w64(valid && !(best <= cap2));
            if (tid == 0 && a.far_lanes != nullptr) a.far_lanes[chunk] = out_of_reach;
        }
    }
}

const char* nn_grid_kernel_name(bool) { return "nn_grid_kernel"; }

// e0 / e1 (both or neither): events that take the kernel's own start and stop timestamps (hipExtLaunchKernelGGL: they ride on the dispatch
// packet's completion signal).  What mi_profile_* times the search with: two hipEventRecord around the launch are two more packets the
// command processor has to work through between kernels -- 8.6 us per ICP step at 1e6 points, where the attached events cost nothing.
hipError_t nn_grid_query(const NnGridView& g, const NnTreeView& t, const GridSearchArgs& a, int fma, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    if (a.n <= 0) return hipSuccess;
    const int n_chunks = (a.n + GRID_BLOCK - 1) / GRID_BLOCK;
    const bool fused = a.state != nullptr;
    const bool helped = fused && a.stats == nullptr && a.split_walks != 0 && a.far_lanes != nullptr;   // two waves per workgroup (nn_grid_kernel<.., 2>)
    const dim3 grid(n_chunks), block(GRID_BLOCK);
    if (fused && (a.order == nullptr || a.far == nullptr || a.rows == nullptr || a.match_slot == nullptr)) return hipErrorInvalidValue;   // (the fused kernel does not test for them)
    const bool timed = e0 != nullptr && e1 != nullptr;
    const bool ext = !fused || a.extend_reach != 0;
#define MI_GRID_LAUNCH_E(F, U, S, E) do { if (timed) hipExtLaunchKernelGGL((nn_grid_kernel<F, U, S, 1, E>), grid, block, 0, s, e0, e1, 0, g, t, a); \
                                          else hipLaunchKernelGGL((nn_grid_kernel<F, U, S, 1, E>), grid, block, 0, s, g, t, a); } while (0)
#define MI_GRID_LAUNCH(F, U, S) do { if (!(U) || ext) MI_GRID_LAUNCH_E(F, U, S, true); else MI_GRID_LAUNCH_E(F, U, S, false); } while (0)
    if (a.stats != nullptr) {          // counting build of the same kernel (mi_profile_search_stats)
        if (fused) { if (fma) MI_GRID_LAUNCH(true, true, true); else MI_GRID_LAUNCH(false, true, true); }
        else { if (fma) MI_GRID_LAUNCH(true, false, true); else MI_GRID_LAUNCH(false, false, true); }
    } else if (helped) {
        const dim3 block2(2 * GRID_BLOCK);
#define MI_GRID_LAUNCH2_E(F, E) do { if (timed) hipExtLaunchKernelGGL((nn_grid_kernel<F, true, false, 2, E>), grid, block2, 0, s, e0, e1, 0, g, t, a); \
                                     else hipLaunchKernelGGL((nn_grid_kernel<F, true, false, 2, E>), grid, block2, 0, s, g, t, a); } while (0)
#define MI_GRID_LAUNCH2(F) do { if (ext) MI_GRID_LAUNCH2_E(F, true); else MI_GRID_LAUNCH2_E(F, false); } while (0)
        if (fma) MI_GRID_LAUNCH2(true); else MI_GRID_LAUNCH2(false);
#undef MI_GRID_LAUNCH2
#undef MI_GRID_LAUNCH2_E
    } else {
        if (fused) { if (fma) MI_GRID_LAUNCH(true, true, false); else MI_GRID_LAUNCH(false, true, false); }
        else { if (fma) MI_GRID_LAUNCH(true, false, false); else MI_GRID_LAUNCH(false, false, false); }
    }
#undef MI_GRID_LAUNCH
#undef MI_GRID_LAUNCH_E
    return hipGetLastError();
}

__global__ void preload_nn_grid_kernel() {}
hipError_t preload_nn_grid()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(preload_nn_grid_kernel));
}

}  // namespace mislam
