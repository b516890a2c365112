// Exact nearest-neighbour search through a box hierarchy over the Morton-sorted fixed cloud (nn_tree.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace mislam {

#ifndef MISLAM_TREE_LEAF
#define MISLAM_TREE_LEAF 8
#endif
// points per leaf.  Measured on MI355X (N = M = 1e6, ms per search early / near convergence; 1e7 early):
//   float4 leaves, 64-byte node records:  4: 2.03 / 1.17 / 36.8    8: 1.84 / 1.08 / 33.3    16: 1.68 / 1.04 / 30.0    32: 1.62 / 1.06 / 28.0
//   compact copies, (node, bound) stack:  8: 1.50 / 0.92 / 27.6    16: 1.35 / 0.89 / 24.4    32: 1.32 / 0.91 / 23.2
//   compact copies, trail + bounds:                     8: 1.25 / 0.71 / 17.8    16: 1.14 / 0.68 / 16.5    32: 1.22 / 0.77 / 17.9
// With at most TREE_NODE_STEPS node visits per round (the default walk; average search of bench.py's 50 iterations at 1e6 / 1e7):
//   32: 0.70 / 6.98 (6 visits)    16: 0.535 / 5.57 (6)    8: 0.487 / 5.06 (5; 0.486 with 4, 0.491 with 6)    4: 0.473 / 5.30 (5)
// -- shorter rounds favour smaller leaves; 8 is within 3 % of the best at both sizes and half the node memory of 4.
constexpr int TREE_LEAF = MISLAM_TREE_LEAF;
constexpr int TREE_MAX_HEIGHT = 24;
// The default walk is the static per-lane kernel (nn_tree_lane_compact_kernel) with a bounded number of node visits per round.
// Measured on MI355X, average search of bench.py's 50 iterations at N = M = 1e6 / 1e7 / 1e5 (ms):
//   node visits per round   unlimited 0.81 / 9.2 / 0.213    8: 0.550    6: 0.535 / 5.57 / 0.156    5: 0.535    4: 0.545    2: 0.66
//   + one contiguous eighth of the moving cloud per XCD (TREE_XCD_CHUNKS)   0.576 / 6.47 / 0.151, FETCH_SIZE 18 MB instead of 53 MB
//   dynamically fetching kernel, 6 visits per round, refill at 24 idle lanes   0.563 / 5.59, FETCH_SIZE 27 MB
//   4-wide kernel, 3 visits per round                                          0.536 / 5.52
constexpr int TREE_NODE_STEPS = 5;             // node visits per round before the wave turns to its leaves (0 = no limit)
constexpr int TREE_LEAF_STEPS = 1;             // leaf scans per round (2 or 3: 0.508 against 0.483 ms at 1e6, 5.28 against 4.99 at 1e7)
// Static kernel, which 256-point chunk a block takes (blocks are dealt to the 8 XCDs round-robin): 0 = chunk b; 1 = every XCD one
// contiguous eighth of the moving cloud; S > 1 = runs of S consecutive chunks per XCD, the XCDs taking neighbouring runs.
// Measured (8-point leaves, 5 visits per round; search ms at 1e6 / 1e7, FETCH_SIZE MB per launch at 1e6):
//   0: 0.483 / 4.99, 62.8    1: 0.513 / 5.81, 19.7 (the eighths are not equally hard)    4: 0.478 / 4.96, 41.9    16: 0.478 / 4.93, 32.3
//   32: 0.479 / 4.93, 29.3    64: 0.492 / 4.94, 29.6    128: 0.492 / 4.94, 33.0
constexpr int TREE_XCD_CHUNKS = 32;
constexpr int TREE_BLOCK_THREADS = 128;        // static kernel: threads per block (the run length above stays in 256-point units)
constexpr bool TREE_DYNAMIC_DEFAULT = false;   // dynamic work fetching (nn_tree_lane_dynamic_kernel) ...
constexpr int TREE_REFILL_MIN = 24;            // ... and the number of finished lanes that triggers a refill
constexpr bool TREE_WIDE_DEFAULT = false;      // the 4-wide walk (nn_tree_wide_kernel)
constexpr int TREE_WIDE_NODE_STEPS = 3;        // its node visits per round
constexpr int TREE_DYNAMIC_PARTS = 8;          // ranges of the moving cloud = XCDs; the work counters are 16 words apart
constexpr int TREE_WORK_COUNTER_WORDS = 16 * TREE_DYNAMIC_PARTS;

struct NnTreeView {
    const float4* pts;                // n_leaves * TREE_LEAF sorted points, w = GLOBAL index bits
    const float4* boxes;              // node i: boxes[2i] = lo, boxes[2i+1] = hi; implicit heap, children of i are 2i+1 and
                                      // 2i+2 (adjacent: one 64-byte record), leaves start at n_pad-1
    int n_pad;                        // leaf count padded to a power of two (padding leaves carry empty boxes)
    int height;                       // log2(n_pad)
    // compact copies read by the default (per-lane) walk: fewer, fuller 16-byte loads per visit
    const float4* pairs;              // internal node p: the boxes of its two children in 3 float4 (48 B instead of 64):
                                      //   (l.lo.x l.lo.y l.lo.z l.hi.x) (l.hi.y l.hi.z r.lo.x r.lo.y) (r.lo.z r.hi.x r.hi.y r.hi.z)
    const uint4* pairs_half;          // the same records in half precision, boxes rounded outwards: 2 x 16 bytes (experimental)
    const float4* leaf_soa;           // leaf f: x[TREE_LEAF], y[TREE_LEAF], z[TREE_LEAF] (3*TREE_LEAF/4 float4; no index word)
    const int* leaf_idx;              // GLOBAL index of sorted slot s (read only for the winner and on exact ties)
    // 4-wide records for the wide walk (nn_tree.hip, tree_pack_quads_kernel): two binary levels per visit
    const float4* quads;
    int quad_levels, quad_parity;     // wide levels above the leaves = ceil(height / 2); height & 1
};

// Morton order of a SoA cloud: order_out[s] = index of the s-th point along the Z-curve of the cloud's bounding box.
struct MortonArgs {
    const float *x, *y, *z;
    int m;
    float* bbox_partials;             // [256][6]
    float* bbox;                      // [6]
    unsigned int *codes_in, *codes_out;
    int *order_in, *order_out;
    void* sort_temp;
    size_t sort_temp_bytes;
};

struct TreeBuildArgs {
    MortonArgs morton;                // over the fixed-cloud shard
    int index_base;                   // global index of point 0
    int n_leaves, n_pad;
    float4* pts;
    float4* boxes;                    // 2 * (2*n_pad - 1) float4
    float4* pairs;                    // 3 * (n_pad - 1) float4
    uint4* pairs_half;                // 2 * (n_pad - 1) uint4, or null
    float4* leaf_soa;                 // n_leaves * 3 * TREE_LEAF / 4 float4
    int* leaf_idx;                    // n_leaves * TREE_LEAF
    float4* quads;                    // 6 * quad_record_count(quad_levels) float4, or null
    int quad_levels, quad_parity;     // ceil(height / 2), height & 1
};

inline size_t quad_record_count(int quad_levels) { return (size_t)(0x55555555u & ((1u << (2 * quad_levels)) - 1u)); }

size_t tree_sort_temp_bytes(int m);
hipError_t morton_order(const MortonArgs& a, hipStream_t s);
hipError_t permute_soa(const float* x, const float* y, const float* z, const int* order, int m, int n_out, float* ox, float* oy,
                       float* oz, hipStream_t s);
hipError_t tree_build(const TreeBuildArgs& a, hipStream_t s);
// R = sources per lane (1 or 2); sources should be Morton-sorted (speed only -- the result never depends on their order)
// name of the kernel nn_tree_query launches for these arguments and the current MISLAM_TREE_* settings (compact copies built)
const char* nn_tree_kernel_name(int n, int R, bool have_counter, int resident_blocks);
// work_counter (one device word, or null) + resident_blocks enable the dynamically fetching form of the default walk.
hipError_t nn_tree_query(const NnTreeView& t, const float* sx, const float* sy, const float* sz, int n, unsigned long long* keys,
                         const int* done_flag, int fma, int R, hipStream_t s, unsigned int* work_counter = nullptr,
                         int resident_blocks = 0);

}  // namespace mislam
