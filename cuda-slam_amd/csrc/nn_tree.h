// Exact nearest-neighbour search through a box hierarchy over the Morton-sorted fixed cloud (nn_tree.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace mislam {

constexpr int TREE_LEAF = 8;          // points per leaf (one leaf = 128 B of float4)
constexpr int TREE_MAX_HEIGHT = 24;

struct NnTreeView {
    const float4* pts;                // n_leaves * TREE_LEAF sorted points, w = GLOBAL index bits
    const float4* box_lo;             // 2*n_pad - 1 nodes, implicit heap: children of i are 2i+1, 2i+2; leaves start at n_pad-1
    const float4* box_hi;
    int n_pad;                        // leaf count padded to a power of two (padding leaves carry empty boxes)
    int height;                       // log2(n_pad)
};

struct TreeBuildArgs {
    const float *tx, *ty, *tz;        // fixed-cloud shard, SoA
    int m;                            // real points
    int index_base;                   // global index of point 0
    int n_leaves, n_pad;
    float* bbox_partials;             // [256][6]
    float* bbox;                      // [6]
    unsigned int *codes_in, *codes_out;
    int *order_in, *order_out;
    void* sort_temp;
    size_t sort_temp_bytes;
    float4* pts;
    float4 *box_lo, *box_hi;
};

size_t tree_sort_temp_bytes(int m);
hipError_t tree_build(const TreeBuildArgs& a, hipStream_t s);
hipError_t nn_tree_query(const NnTreeView& t, const float* sx, const float* sy, const float* sz, int n, unsigned long long* keys,
                         const int* done_flag, int fma, hipStream_t s);

}  // namespace mislam
