// K1t -- EXACT nearest-neighbour search through a bounding-box hierarchy over the Morton-sorted fixed cloud.
//
// Same contract as the brute-force K1 (nn_kernel.hip) and therefore as FindCorrespondences (cudacommon.cu:57-77) /
// common.cpp:446-462: idx[i] = argmin_j |after[j] - before[i]|^2 under strict '<' with the lowest index winning ties, the
// distance evaluated with the same fp32 operation sequence.  The result is IDENTICAL to brute force, bit for bit, because
//   * a leaf point is accepted iff (d, j) is lexicographically smaller than the running (best, bidx) -- the order in which
//     candidates are met does not matter for a lexicographic minimum;
//   * a subtree is skipped only if its box bound lb is STRICTLY greater than best, where lb is computed with the very same
//     rounded operations as a distance: for every point q of the box and every axis, |fl(q - s)| >= fl(e) with
//     e = max(lo - s, s - hi, 0) (rounding is monotonic), hence fl-by-fl lb <= d(q): a skipped point can neither win nor tie.
//
// Build (once per fixed cloud / shard; the fixed cloud does not move during ICP): bounding box -> 30-bit Morton codes ->
// radix sort (rocPRIM device primitive; one-time index build, not the per-iteration path) -> leaves of 8 consecutive
// points as float4 (x, y, z, global index bits) -> implicit binary heap of boxes over the leaves (padded to a power of
// two with empty boxes).
// Query: one lane per source point, depth-first, nearer child first, per-lane stack in LDS; starts from the key already
// posted for the point (the previous ICP iteration's match under the new transform), so late iterations mostly verify.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "kernels.h"
#include "nn_tree.h"

namespace mislam {

// ---------------------------------------------------------------------------------------------------------------
// build
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
    codes[j] = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
    order[j] = j;
}

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

__global__ __launch_bounds__(256) void tree_leaf_box_kernel(const float4* __restrict__ pts, int n_leaves, int n_pad,
                                                            float4* __restrict__ box_lo, float4* __restrict__ box_hi)
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
    const int node = n_pad - 1 + leaf;
    box_lo[node] = make_float4(lo[0], lo[1], lo[2], 0.f);
    box_hi[node] = make_float4(hi[0], hi[1], hi[2], 0.f);
}

// nodes [first, first + count) of one level: box = union of the two children
__global__ __launch_bounds__(256) void tree_level_kernel(int first, int count, float4* __restrict__ box_lo, float4* __restrict__ box_hi)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int node = first + i, l = 2 * node + 1, r = l + 1;
    const float4 a = box_lo[l], b = box_lo[r], c = box_hi[l], d = box_hi[r];
    box_lo[node] = make_float4(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), 0.f);
    box_hi[node] = make_float4(fmaxf(c.x, d.x), fmaxf(c.y, d.y), fmaxf(c.z, d.z), 0.f);
}

size_t tree_sort_temp_bytes(int m)
{
    size_t bytes = 0;
    unsigned int* k = nullptr;
    int* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)m, 0u, 30u, (hipStream_t)0, false);
    return bytes;
}

hipError_t tree_build(const TreeBuildArgs& a, hipStream_t s)
{
    const int m = a.m;
    const int blocks = (m + 255) / 256;
    const int rb = blocks < 256 ? blocks : 256;
    hipLaunchKernelGGL(tree_bbox_partial_kernel, dim3(rb), dim3(256), 0, s, a.tx, a.ty, a.tz, m, a.bbox_partials);
    hipLaunchKernelGGL(tree_bbox_final_kernel, dim3(1), dim3(64), 0, s, a.bbox_partials, rb, a.bbox);
    hipLaunchKernelGGL(tree_morton_kernel, dim3(blocks), dim3(256), 0, s, a.tx, a.ty, a.tz, m, a.bbox, a.codes_in, a.order_in);
    size_t temp = a.sort_temp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(a.sort_temp, temp, a.codes_in, a.codes_out, a.order_in, a.order_out, (size_t)m, 0u, 30u, s, false);
    if (e != hipSuccess) return e;
    const int n_slots = a.n_leaves * TREE_LEAF;
    hipLaunchKernelGGL(tree_gather_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, s, a.tx, a.ty, a.tz, a.order_out, m, n_slots,
                       a.index_base, a.pts);
    hipLaunchKernelGGL(tree_leaf_box_kernel, dim3((a.n_pad + 255) / 256), dim3(256), 0, s, a.pts, a.n_leaves, a.n_pad, a.box_lo, a.box_hi);
    for (int count = a.n_pad / 2; count >= 1; count /= 2) {   // levels bottom-up: nodes [count-1, 2*count-1)
        hipLaunchKernelGGL(tree_level_kernel, dim3((count + 255) / 256), dim3(256), 0, s, count - 1, count, a.box_lo, a.box_hi);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// query
// ---------------------------------------------------------------------------------------------------------------
template <bool FMA>
__device__ __forceinline__ float sq3(float dx, float dy, float dz)
{
    if constexpr (FMA) return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    else return (dx * dx + dy * dy) + dz * dz;
}

template <bool FMA>
__device__ __forceinline__ float box_bound(const float4 lo, const float4 hi, float sx, float sy, float sz)
{
    const float ex = fmaxf(fmaxf(lo.x - sx, sx - hi.x), 0.f);
    const float ey = fmaxf(fmaxf(lo.y - sy, sy - hi.y), 0.f);
    const float ez = fmaxf(fmaxf(lo.z - sz, sz - hi.z), 0.f);
    return sq3<FMA>(ex, ey, ez);
}

// One lane per source point.  Dynamic LDS: stack_depth x 256 x 8 bytes (node id + bound per entry, column per lane).
template <bool FMA>
__global__ __launch_bounds__(256) void nn_tree_query_kernel(NnTreeView t, const float* __restrict__ sx, const float* __restrict__ sy,
                                                            const float* __restrict__ sz, int n, unsigned long long* __restrict__ keys,
                                                            const int* __restrict__ done_flag, int stack_depth)
{
    if (done_flag != nullptr && *done_flag != 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* st_node = reinterpret_cast<int*>(smem);
    float* st_lb = reinterpret_cast<float*>(smem + (size_t)stack_depth * 256 * sizeof(int));

    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float px = sx[i], py = sy[i], pz = sz[i];

    // starting candidate: the key already posted for this source (KEY_INIT -> none)
    const unsigned long long k0 = keys[i];
    unsigned int hi0 = (unsigned int)(k0 >> 32);
    float best = hi0 < 0x7f800000u ? __uint_as_float(hi0) : __builtin_inff();
    // no candidate yet: (inf, 0) -- no index is < 0, so a point at overflowed distance +inf is never accepted, exactly like
    // the brute-force kernel's strict 'd < inf'
    unsigned int bidx = hi0 < 0x7f800000u ? (unsigned int)(k0 & 0xffffffffull) : 0u;

    const int first_leaf = t.n_pad - 1;
    int sp = 0;
    int node = 0;
    float node_lb = box_bound<FMA>(t.box_lo[0], t.box_hi[0], px, py, pz);
    bool have = node_lb <= best && node_lb < __builtin_inff();
    while (true) {
        if (!have) {
            // pop until an entry that can still matter
            bool found = false;
            while (sp > 0) {
                sp--;
                const float lb = st_lb[sp * 256 + threadIdx.x];
                if (lb <= best) { node = st_node[sp * 256 + threadIdx.x]; found = true; break; }
            }
            if (!found) break;
        }
        have = false;
        if (node >= first_leaf) {
            const float4* __restrict__ lp = t.pts + (size_t)(node - first_leaf) * TREE_LEAF;
#pragma unroll
            for (int k = 0; k < TREE_LEAF; k++) {
                const float4 q = lp[k];
                const float d = sq3<FMA>(q.x - px, q.y - py, q.z - pz);
                const unsigned int j = (unsigned int)__float_as_int(q.w);
                if (d < best || (d == best && j < bidx)) { best = d; bidx = j; }
            }
        } else {
            const int l = 2 * node + 1, r = l + 1;
            const float lbl = box_bound<FMA>(t.box_lo[l], t.box_hi[l], px, py, pz);
            const float lbr = box_bound<FMA>(t.box_lo[r], t.box_hi[r], px, py, pz);
            const bool left_near = lbl <= lbr;
            const int near = left_near ? l : r, far = left_near ? r : l;
            const float lbn = left_near ? lbl : lbr, lbf = left_near ? lbr : lbl;
            if (lbf <= best && lbf < __builtin_inff()) {
                st_node[sp * 256 + threadIdx.x] = far;
                st_lb[sp * 256 + threadIdx.x] = lbf;
                sp++;
            }
            if (lbn <= best && lbn < __builtin_inff()) { node = near; have = true; }
        }
    }
    if (best < __builtin_inff()) keys[i] = ((unsigned long long)__float_as_uint(best) << 32) | bidx;
}

hipError_t nn_tree_query(const NnTreeView& t, const float* sx, const float* sy, const float* sz, int n, unsigned long long* keys,
                         const int* done_flag, int fma, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const int depth = t.height + 2;
    const size_t lds = (size_t)depth * 256 * 8;
    dim3 grid((n + 255) / 256), block(256);
    if (fma) hipLaunchKernelGGL(nn_tree_query_kernel<true>, grid, block, lds, s, t, sx, sy, sz, n, keys, done_flag, depth);
    else hipLaunchKernelGGL(nn_tree_query_kernel<false>, grid, block, lds, s, t, sx, sy, sz, n, keys, done_flag, depth);
    return hipGetLastError();
}

}  // namespace mislam
