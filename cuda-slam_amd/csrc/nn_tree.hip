// K1t -- EXACT nearest-neighbour search through a bounding-box hierarchy over the Morton-sorted fixed cloud.
//
// Same contract as the every-pair K1 (nn_kernel.hip) and therefore as FindCorrespondences (cudacommon.cu:57-77) /
// common.cpp:446-462; the result is IDENTICAL to the every-pair search, bit for bit (the two rules in nn_walk.hpp).
//
// Build (once per fixed cloud / shard -- the fixed cloud does not move during ICP): bounding box -> 30-bit Morton codes ->
// radix sort (radix_sort.hip, own kernels; one-time index build at load) -> leaves of TREE_LEAF
// consecutive points -> implicit binary heap of boxes over the leaves, padded to a power of two with empty boxes.  The walk
// reads compact copies: a node's box as six floats (eight consecutive boxes per step), a leaf as x[8] y[8] z[8], the global
// indices in a side array that is read only for the winner and on exact ties.
//
// Query: one wave per 64 moving points, ONE cooperative walk per wave (nn_walk.hpp tree_walk_wide).  Round 1 walked per lane: a
// chain of a few hundred dependent, divergent 16-byte loads per lane (~325 L1 line accesses per query, 24.7 of 64 lanes active,
// 0.49 ms per search at N = M = 1e6).  Since round 2 the default search is the cell grid (nn_grid.hip) and the walk is its
// in-kernel fallback for the lanes the grid cannot serve cheaply (no starting candidate yet, far outside the fixed cloud,
// crowded cells); MI_NN_TREE still runs it for every point.  The per-lane walk forms of round 1 are gone; their measurements stay
// in DESIGN.md.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "nn_tree.h"
#include "nn_walk.hpp"

namespace mislam {
// ---------------------------------------------------------------------------------------------------------------
// Morton order of a SoA cloud (used for the fixed cloud's leaves and for the moving cloud's lane grouping)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tree_bbox_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                const float* __restrict__ z, int m, float* __restrict__ partials)
{
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int j = blockIdx.x * 256 + threadIdx.x; j < m; j += gridDim.x * 256) {
        const float p[3] = {x[j], y[j], z[j]};
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], p[a]); hi[a] = fmaxf(hi[a], p[a]); }
    }
    __shared__ float s[6][256];
    for (int a = 0; a < 3; a++) { s[a][threadIdx.x] = lo[a]; s[3 + a][threadIdx.x] = hi[a]; }
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int a = 0; a < 3; a++) {
                s[a][threadIdx.x] = fminf(s[a][threadIdx.x], s[a][threadIdx.x + w]);
                s[3 + a][threadIdx.x] = fmaxf(s[3 + a][threadIdx.x], s[3 + a][threadIdx.x + w]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) partials[blockIdx.x * 6 + threadIdx.x] = s[threadIdx.x][0];
}

__global__ void tree_bbox_final_kernel(const float* __restrict__ partials, int nblocks, float* __restrict__ bbox)
{
    const int a = threadIdx.x;
    if (a >= 6) return;
    float v = partials[a];
    for (int b = 1; b < nblocks; b++) v = a < 3 ? fminf(v, partials[b * 6 + a]) : fmaxf(v, partials[b * 6 + a]);
    bbox[a] = v;
}

__device__ __forceinline__ unsigned int spread10(unsigned int v)   // 10 bits -> every third bit
{
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// Position of a 10-bit lattice point along the HILBERT curve of the bounding box (Skilling's transpose form, "Programming the
// Hilbert curve", 2004).  Unlike the Z-order the Hilbert curve has no jumps: any run of consecutive points is spatially compact.
// That is what the cooperative walk wants from the moving cloud -- a wave's 64 points walk TOGETHER, and a wave that straddled a
// jump of the Z-curve paid for two distant groups (its walk, the longest of the launch, set the launch's duration) -- and what
// the implicit heap wants from the fixed cloud (every subtree is a run of consecutive leaves).
__device__ __forceinline__ unsigned int hilbert30(unsigned int x, unsigned int y, unsigned int z)
{
    unsigned int X[3] = {x, y, z};
    for (unsigned int Q = 1u << 9; Q > 1u; Q >>= 1) {
        const unsigned int P = Q - 1u;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (X[i] & Q) X[0] ^= P;
            else { const unsigned int t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    unsigned int t = 0;
    for (unsigned int Q = 1u << 9; Q > 1u; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1u;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    return (spread10(X[0]) << 2) | (spread10(X[1]) << 1) | spread10(X[2]);
}

__global__ __launch_bounds__(256) void tree_morton_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ z, int m, const float* __restrict__ bbox,
                                                          unsigned int* __restrict__ codes, int* __restrict__ order)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const float p[3] = {x[j], y[j], z[j]};
    unsigned int q[3];
    for (int a = 0; a < 3; a++) {
        const float ext = bbox[3 + a] - bbox[a];
        float u = ext > 0.f ? (p[a] - bbox[a]) / ext : 0.f;
        u = fminf(fmaxf(u * 1024.f, 0.f), 1023.f);
        q[a] = (unsigned int)u;
    }
#ifdef MISLAM_DEV_MORTON
    codes[j] = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
#else
    codes[j] = hilbert30(q[0], q[1], q[2]);
#endif
    order[j] = j;
}

size_t tree_sort_temp_bytes(int m) { return radix_sort_temp_bytes(m); }

hipError_t cloud_bbox(const float* x, const float* y, const float* z, int m, float* partials, float* bbox, hipStream_t s)
{
    const int blocks = (m + 255) / 256;
    const int rb = blocks < 256 ? blocks : 256;
    hipLaunchKernelGGL(tree_bbox_partial_kernel, dim3(rb), dim3(256), 0, s, x, y, z, m, partials);
    hipLaunchKernelGGL(tree_bbox_final_kernel, dim3(1), dim3(64), 0, s, partials, rb, bbox);
    return hipGetLastError();
}

hipError_t morton_order(const MortonArgs& a, hipStream_t s)
{
    const int m = a.m;
    const int blocks = (m + 255) / 256;
    hipError_t be = cloud_bbox(a.x, a.y, a.z, m, a.bbox_partials, a.bbox, s);
    if (be != hipSuccess) return be;
    hipLaunchKernelGGL(tree_morton_kernel, dim3(blocks), dim3(256), 0, s, a.x, a.y, a.z, m, a.bbox, a.codes_in, a.order_in);
    // order_in doubles as the ping-pong partner of order_out (three passes: the result lands in the *_out arrays)
    return radix_sort_pairs_u32(a.sort_temp, a.codes_in, a.codes_out, a.order_in, a.order_out, m, 30, s);
}

// out[s] = in[order[min(s, m-1)]] for s < n_out (tail replicates the last sorted point)
__global__ __launch_bounds__(256) void permute_soa_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ z, const int* __restrict__ order, int m, int n_out,
                                                          float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_out) return;
    const int j = order[s < m ? s : m - 1];
    ox[s] = x[j]; oy[s] = y[j]; oz[s] = z[j];
}

hipError_t permute_soa(const float* x, const float* y, const float* z, const int* order, int m, int n_out, float* ox, float* oy,
                       float* oz, hipStream_t s)
{
    if (n_out <= 0) return hipSuccess;
    hipLaunchKernelGGL(permute_soa_kernel, dim3((n_out + 255) / 256), dim3(256), 0, s, x, y, z, order, m, n_out, ox, oy, oz);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// hierarchy build
// ---------------------------------------------------------------------------------------------------------------
// sorted slot s <- point order[s]; slots past the last real point replicate it (same coordinates AND same index: a no-op
// for a lexicographic minimum)
__global__ __launch_bounds__(256) void tree_gather_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ z, const int* __restrict__ order, int m,
                                                          int n_slots, int index_base, float4* __restrict__ pts)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_slots) return;
    const int j = order[s < m ? s : m - 1];
    pts[s] = make_float4(x[j], y[j], z[j], __int_as_float(j + index_base));
}

// boxes[2*node] = lo, boxes[2*node+1] = hi
__global__ __launch_bounds__(256) void tree_leaf_box_kernel(const float4* __restrict__ pts, int n_leaves, int n_pad, float4* __restrict__ boxes)
{
    const int leaf = blockIdx.x * 256 + threadIdx.x;
    if (leaf >= n_pad) return;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    if (leaf < n_leaves) {
        for (int k = 0; k < TREE_LEAF; k++) {
            const float4 p = pts[(size_t)leaf * TREE_LEAF + k];
            lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
            hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
        }
    }
    const size_t node = (size_t)(n_pad - 1 + leaf);
    boxes[2 * node] = make_float4(lo[0], lo[1], lo[2], 0.f);
    boxes[2 * node + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
}

// nodes [first, first + count) of one level: box = union of the two children
__global__ __launch_bounds__(256) void tree_level_kernel(int first, int count, float4* __restrict__ boxes)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const size_t node = (size_t)(first + i), l = 2 * node + 1, r = l + 1;
    const float4 a = boxes[2 * l], c = boxes[2 * l + 1], b = boxes[2 * r], d = boxes[2 * r + 1];
    boxes[2 * node] = make_float4(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), 0.f);
    boxes[2 * node + 1] = make_float4(fmaxf(c.x, d.x), fmaxf(c.y, d.y), fmaxf(c.z, d.z), 0.f);
}

// node boxes for the wide walk, SIBLINGS INTERLEAVED (NnTreeView::boxes6): pair q holds nodes 2q-1 and 2q component by component; the tail
// padding (and the unused first half of pair 0) gets empty boxes
__global__ __launch_bounds__(256) void tree_pack_boxes6_kernel(const float4* __restrict__ boxes, int n_nodes, int n_out, float* __restrict__ boxes6)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    float4 lo = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f);
    float4 hi = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), 0.f);
    if (i == 0) {                                              // (the half of pair 0 that is nobody's)
        float* e = boxes6;
        e[0] = lo.x; e[2] = lo.y; e[4] = lo.z; e[6] = hi.x; e[8] = hi.y; e[10] = hi.z;
    }
    if (i < n_nodes) { lo = boxes[2 * (size_t)i]; hi = boxes[2 * (size_t)i + 1]; }
    float* o = boxes6 + tree_box_offset(i);
    o[0] = lo.x; o[2] = lo.y; o[4] = lo.z; o[6] = hi.x; o[8] = hi.y; o[10] = hi.z;
}

__global__ __launch_bounds__(256) void tree_pack_leaves_kernel(const float4* __restrict__ pts, int n_slots, float* __restrict__ soa,
                                                               int* __restrict__ idx)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_slots) return;
    const float4 q = pts[s];
    const int leaf = s / TREE_LEAF, k = s % TREE_LEAF;
    float* base = soa + (size_t)leaf * 3 * TREE_LEAF;
    base[k] = q.x; base[TREE_LEAF + k] = q.y; base[2 * TREE_LEAF + k] = q.z;
    idx[s] = __float_as_int(q.w);
}

hipError_t tree_build(const TreeBuildArgs& a, hipStream_t s)
{
    hipError_t e = morton_order(a.morton, s);
    if (e != hipSuccess) return e;
    const int m = a.morton.m;
    const int n_slots = a.n_leaves * TREE_LEAF;
    hipLaunchKernelGGL(tree_gather_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, s, a.morton.x, a.morton.y, a.morton.z,
                       a.morton.order_out, m, n_slots, a.index_base, a.pts);
    hipLaunchKernelGGL(tree_leaf_box_kernel, dim3((a.n_pad + 255) / 256), dim3(256), 0, s, a.pts, a.n_leaves, a.n_pad, a.boxes);
    for (int count = a.n_pad / 2; count >= 1; count /= 2)   // levels bottom-up: nodes [count-1, 2*count-1)
        hipLaunchKernelGGL(tree_level_kernel, dim3((count + 255) / 256), dim3(256), 0, s, count - 1, count, a.boxes);
    const int n_nodes = 2 * a.n_pad - 1;
    hipLaunchKernelGGL(tree_pack_boxes6_kernel, dim3((n_nodes + 8 + 255) / 256), dim3(256), 0, s, a.boxes, n_nodes, n_nodes + 8, a.boxes6);
    hipLaunchKernelGGL(tree_pack_leaves_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, s, a.pts, n_slots, reinterpret_cast<float*>(a.leaf_soa),
                       a.leaf_idx);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// query: one lane per moving point, one wave per workgroup (nn_walk.hpp tree_walk_wave)
// ---------------------------------------------------------------------------------------------------------------
template <bool FMA>
__global__ __launch_bounds__(TREE_BLOCK_THREADS) void nn_tree_kernel(NnTreeView t, const float* __restrict__ sx, const float* __restrict__ sy,
                                                                          const float* __restrict__ sz, int n, unsigned long long* __restrict__ keys,
                                                                          const int* __restrict__ done_flag)
{
    if (done_flag != nullptr && *done_flag != 0) return;
    const unsigned int chunk = xcd_chunk(blockIdx.x, gridDim.x, TREE_XCD_CHUNKS * (256 / TREE_BLOCK_THREADS));
    const int i = (int)(chunk * TREE_BLOCK_THREADS + threadIdx.x);
    if (i >= n) return;
    const float p[3] = {sx[i], sy[i], sz[i]};
    float best;
    unsigned int bidx;
    unpack_start(keys[i], best, bidx);
    unsigned int n_nodes = 0u, n_leaves = 0u;
    tree_walk_wide<FMA, false>(t, p, best, bidx, n_nodes, n_leaves);
    if (best < __builtin_inff()) keys[i] = ((unsigned long long)__float_as_uint(best) << 32) | bidx;
}

hipError_t nn_tree_query(const NnTreeView& t, const float* sx, const float* sy, const float* sz, int n, unsigned long long* keys,
                         const int* done_flag, int fma, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const dim3 grid((n + TREE_BLOCK_THREADS - 1) / TREE_BLOCK_THREADS), block(TREE_BLOCK_THREADS);
    if (fma) hipLaunchKernelGGL(nn_tree_kernel<true>, grid, block, 0, s, t, sx, sy, sz, n, keys, done_flag);
    else hipLaunchKernelGGL(nn_tree_kernel<false>, grid, block, 0, s, t, sx, sy, sz, n, keys, done_flag);
    return hipGetLastError();
}

// Touching one kernel of this translation unit makes the runtime load its code object now (mi_ctx_create) instead of at the
// first launch inside a registration call (deferred loading: 5-16 ms per object, once).
__global__ void preload_nn_tree_kernel() {}
hipError_t preload_nn_tree()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(preload_nn_tree_kernel));
}

}  // namespace mislam
