// Exact nearest-neighbour search through a box hierarchy over the Morton-sorted fixed cloud (nn_tree.hip).  Since round 2 the
// hierarchy is the FALLBACK of the cell-grid search (nn_grid.hip): lanes whose search radius or candidate count is too large for
// the grid walk the hierarchy instead, inside the same kernel.  MI_NN_TREE still runs it alone.
#pragma once
#include <hip/hip_runtime.h>

namespace mislam {

// points per leaf.  Measured on MI355X in round 1 (bounded rounds; average search of bench.py's 50 iterations at 1e6 / 1e7, ms):
//   32: 0.70 / 6.98    16: 0.535 / 5.57    8: 0.487 / 5.06    4: 0.473 / 5.30
#ifndef MISLAM_TREE_LEAF
#define MISLAM_TREE_LEAF 8
#endif
constexpr int TREE_LEAF = MISLAM_TREE_LEAF;
constexpr int TREE_MAX_HEIGHT = 24;
// which 256-point chunk of the Morton-sorted moving cloud a workgroup takes: runs of TREE_XCD_CHUNKS consecutive chunks per XCD
// (workgroups are dealt to the 8 XCDs round-robin), the XCDs taking neighbouring runs -- an XCD then walks contiguous stretches
// of the Z-curve and its L2 fetches each part of the index about once (profiles/r01_k1t_xcd_runs.log)
constexpr int TREE_XCD_CHUNKS = 32;
constexpr int TREE_BLOCK_THREADS = 64;         // one wave per workgroup: a walk is wave-wide, nothing is shared between waves, no LDS

// float offset of node i's lo.x in NnTreeView::boxes6 (its other components follow two floats apart): odd i is the first of its pair, even
// i (and the root, alone in pair 0) the second
__host__ __device__ inline size_t tree_box_offset(int i) { return (size_t)((i + 1) >> 1) * 12 + (size_t)((i & 1) ^ 1); }

struct NnTreeView {
    int n_pad;                        // leaf count padded to a power of two (padding leaves carry empty boxes)
    int height;                       // log2(n_pad)
    int n_leaves;                     // real leaves: a node is empty iff the first leaf below it is >= n_leaves
    const float* boxes6;              // node boxes, implicit heap (children of i are 2i+1 and 2i+2, leaves start at n_pad-1, so the 2^k
                                      // descendants k levels below a node are contiguous), SIBLINGS INTERLEAVED: pair q = nodes 2q-1, 2q as
                                      // lo.x lo.x' lo.y lo.y' lo.z lo.z' hi.x hi.x' hi.y hi.y' hi.z hi.z' (48 bytes; tree_box_offset) -- the
                                      // walk bounds two boxes per packed instruction straight out of its scalar loads; a step reads four
                                      // consecutive pairs (+ padding pairs at the end: a step always reads eight boxes)
    const float4* leaf_soa;           // leaf f: x[TREE_LEAF], y[TREE_LEAF], z[TREE_LEAF] (3*TREE_LEAF/4 float4; no index word)
    const int* leaf_idx;              // GLOBAL index of sorted slot s (read only for the winner and on exact ties)
};

// Morton order of a SoA cloud: order_out[s] = index of the s-th point along the Z-curve of the cloud's bounding box.
struct MortonArgs {
    const float *x, *y, *z;
    int m;
    float* bbox_partials;             // [256][6]
    float* bbox;                      // [6]: lo xyz, hi xyz
    unsigned int *codes_in, *codes_out;
    int *order_in, *order_out;
    void* sort_temp;
    size_t sort_temp_bytes;
};

struct TreeBuildArgs {
    MortonArgs morton;                // over the fixed-cloud shard
    int index_base;                   // global index of point 0
    int n_leaves, n_pad;
    float4* pts;                      // scratch: n_leaves * TREE_LEAF sorted points (x, y, z, global-index bits)
    float4* boxes;                    // scratch: 2 * (2*n_pad - 1) float4 (lo, hi per node)
    float* boxes6;                    // 12 floats per pair of nodes: 12 * (n_pad + 5) floats
    float4* leaf_soa;                 // n_leaves * 3 * TREE_LEAF / 4 float4
    int* leaf_idx;                    // n_leaves * TREE_LEAF
};

size_t tree_sort_temp_bytes(int m);
// bounding box of a SoA cloud into bbox[6] (two small launches; partials = 256*6 floats of scratch)
hipError_t cloud_bbox(const float* x, const float* y, const float* z, int m, float* partials, float* bbox, hipStream_t s);
hipError_t morton_order(const MortonArgs& a, hipStream_t s);
hipError_t permute_soa(const float* x, const float* y, const float* z, const int* order, int m, int n_out, float* ox, float* oy,
                       float* oz, hipStream_t s);
hipError_t tree_build(const TreeBuildArgs& a, hipStream_t s);
// one lane per moving point (sources should be Morton-sorted: speed only -- the result never depends on their order)
hipError_t nn_tree_query(const NnTreeView& t, const float* sx, const float* sy, const float* sz, int n, unsigned long long* keys,
                         const int* done_flag, int fma, hipStream_t s);

}  // namespace mislam
