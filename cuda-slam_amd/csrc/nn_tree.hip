// K1t -- EXACT nearest-neighbour search through a bounding-box hierarchy over the Morton-sorted fixed cloud.
//
// Same contract as the every-pair K1 (nn_kernel.hip) and therefore as FindCorrespondences (cudacommon.cu:57-77) /
// common.cpp:446-462: idx[i] = argmin_j |after[j] - before[i]|^2 under strict '<' with the lowest index winning ties, the
// distance evaluated with the same fp32 operation sequence.  The result is IDENTICAL to the every-pair search, bit for bit:
//   * a candidate is accepted iff (d, j) is lexicographically smaller than the running (best, bidx) -- the order in which
//     candidates are met does not matter for a lexicographic minimum;
//   * a subtree is skipped only if a lower bound lb of its box is STRICTLY greater than the bound it is compared with, and lb is
//     computed with the very same rounded operations as a distance.  For a point q in the node box [lo,hi] and a source s in
//     the group box [glo,ghi], per axis q - s >= lo - ghi and s - q >= glo - hi in real arithmetic; rounding is monotonic, so
//     |fl(q - s)| >= fl(gap) with gap = max(lo - ghi, glo - hi, 0), and squaring / summing in the distance's own order keeps
//     the inequality: fl-by-fl lb <= d(q, s).  A skipped point can neither win nor tie.  (A single source is the
//     degenerate group glo = ghi = s.)
//
// Build (once per fixed cloud / shard -- the fixed cloud does not move during ICP): bounding box -> 30-bit Morton codes ->
// radix sort (rocPRIM device primitive; one-time index build, not the per-iteration path) -> leaves of TREE_LEAF (16) consecutive
// points as float4 (x, y, z, global-index bits) -> implicit binary heap of boxes over the leaves, padded to a power of two
// with empty boxes.  Node i's children are 2i+1 and 2i+2 and their boxes are ADJACENT in memory: one 64-byte record.
//
// The moving cloud is Morton-sorted once at load, so the 64 lanes of a wave hold 64 spatial neighbours.  Two query forms,
// both exact (the result never depends on which one runs):
//   * per-lane (default, nn_tree_lane_kernel): each lane walks the hierarchy for its own source, nearer child first,
//     per-lane stack in LDS, pruning with the lane's OWN best.  Neighbouring lanes take similar paths, so most node loads
//     of a wave coalesce.  Measured on MI355X, N = M = 1e6 synthetic: 1.8 ms per search in early ICP iterations
//     (radius 0.44, ~360 points inside the search sphere), 1.1 ms near convergence; 2.7 ms without the source sort.
//   * per-lane, stackless (nn_tree_trail_kernel, MISLAM_TREE_R=-1): the heap numbering replaces the stack by a 32-bit trail of
//     pending levels -- no LDS, full occupancy, but one extra box load per pending sibling: 1.84 / 1.09 ms, i.e. the walk is
//     bound by vector-memory request throughput, not by occupancy.
//   * wave-cooperative (nn_tree_wave_kernel, MISLAM_TREE_R=1|2): the wave walks the hierarchy ONCE for the whole group; every
//     address is wave-uniform, so boxes and leaf points arrive through SCALAR loads and feed the VALU as SGPR operands (like
//     K1), control flow is uniform, and the stack lives across the lanes of one VGPR (v_writelane / v_readlane indexed by
//     the scalar stack pointer) -- no LDS.  It prunes against the group's WORST running best, so it evaluates
//     ~(s + 2r)^3 / (4.2 r^3) times more pairs than the per-lane form (s = extent of the 64 neighbours, r = search radius),
//     and a group that straddles a jump of the Z-curve has a box as large as the cloud and degenerates to the every-pair
//     search for that wave.  Measured: 77 ms early / 27 ms late at N = 1e6 -- kept as a documented, tested alternative, not
//     the default.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "kernels.h"
#include "nn_tree.h"

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

size_t tree_sort_temp_bytes(int m)
{
    size_t bytes = 0;
    unsigned int* k = nullptr;
    int* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)m, 0u, 30u, (hipStream_t)0, false);
    return bytes;
}

hipError_t morton_order(const MortonArgs& a, hipStream_t s)
{
    const int m = a.m;
    const int blocks = (m + 255) / 256;
    const int rb = blocks < 256 ? blocks : 256;
    hipLaunchKernelGGL(tree_bbox_partial_kernel, dim3(rb), dim3(256), 0, s, a.x, a.y, a.z, m, a.bbox_partials);
    hipLaunchKernelGGL(tree_bbox_final_kernel, dim3(1), dim3(64), 0, s, a.bbox_partials, rb, a.bbox);
    hipLaunchKernelGGL(tree_morton_kernel, dim3(blocks), dim3(256), 0, s, a.x, a.y, a.z, m, a.bbox, a.codes_in, a.order_in);
    size_t temp = a.sort_temp_bytes;
    return rocprim::radix_sort_pairs(a.sort_temp, temp, a.codes_in, a.codes_out, a.order_in, a.order_out, (size_t)m, 0u, 30u, s, false);
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

// compact copies for the per-lane walk (layouts: NnTreeView)
__global__ __launch_bounds__(256) void tree_pack_pairs_kernel(const float4* __restrict__ boxes, int n_internal, float4* __restrict__ pairs)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_internal) return;
    const int l = 2 * p + 1;
    const float4 llo = boxes[2 * (size_t)l], lhi = boxes[2 * (size_t)l + 1], rlo = boxes[2 * (size_t)l + 2], rhi = boxes[2 * (size_t)l + 3];
    pairs[3 * (size_t)p] = make_float4(llo.x, llo.y, llo.z, lhi.x);
    pairs[3 * (size_t)p + 1] = make_float4(lhi.y, lhi.z, rlo.x, rlo.y);
    pairs[3 * (size_t)p + 2] = make_float4(rlo.z, rhi.x, rhi.y, rhi.z);
}

// 4-wide records for the wide walk: the same hierarchy read two binary levels at a time.  A wide node at wide level k, position
// j stands for the binary node at binary level 2k - parity, position j (parity = height & 1; with an odd height the wide root
// has two children and two empty boxes); its record holds the boxes of its four wide children, one coordinate of all four per
// float4: (lo.x x4)(lo.y x4)(lo.z x4)(hi.x x4)(hi.y x4)(hi.z x4) = 96 bytes.  Wide level k starts at record (4^k - 1) / 3.
__host__ __device__ __forceinline__ unsigned int quad_level_offset(int k) { return 0x55555555u & ((1u << (2 * k)) - 1u); }

__global__ __launch_bounds__(256) void tree_pack_quads_kernel(const float4* __restrict__ boxes, int quad_levels, int parity, float4* __restrict__ quads)
{
    const unsigned int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= quad_level_offset(quad_levels)) return;
    int k = 0;
    while (quad_level_offset(k + 1) <= q) k++;
    const unsigned int pos = q - quad_level_offset(k);
    const int child_binary_level = 2 * (k + 1) - parity;
    float lo[3][4], hi[3][4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const unsigned int cp = 4u * pos + c;
        float4 l = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f);
        float4 h = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), 0.f);
        if (cp < (1u << child_binary_level)) {
            const size_t b = (size_t)((1u << child_binary_level) - 1u) + cp;
            l = boxes[2 * b];
            h = boxes[2 * b + 1];
        }
        lo[0][c] = l.x; lo[1][c] = l.y; lo[2][c] = l.z;
        hi[0][c] = h.x; hi[1][c] = h.y; hi[2][c] = h.z;
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        quads[6 * (size_t)q + a] = make_float4(lo[a][0], lo[a][1], lo[a][2], lo[a][3]);
        quads[6 * (size_t)q + 3 + a] = make_float4(hi[a][0], hi[a][1], hi[a][2], hi[a][3]);
    }
}

// Sibling-box records in half precision: 12 values in 24 of 32 bytes (2 loads instead of 3).  A value w of axis a is stored as
// h ~ (w - c_a) / s_a (c, s: centre and half-extent of the root box, so |h| <= 1 and the grid is 2^-11 of the extent at
// worst) and decoded as fma(float(h), s_a, c_a).  Lower corners are rounded DOWN and upper corners UP until the DECODED value
// is on the outer side of w, so a decoded box always contains the exact one: the walk's bound stays a lower bound (the same
// monotonicity argument as for the fp32 boxes), it only prunes a hair less.
__device__ __forceinline__ void tree_half_frame(const float4 root_lo, const float4 root_hi, float c[3], float s[3])
{
    const float lo[3] = {root_lo.x, root_lo.y, root_lo.z}, hi[3] = {root_hi.x, root_hi.y, root_hi.z};
    for (int a = 0; a < 3; a++) {
        c[a] = 0.5f * (lo[a] + hi[a]);
        const float e = 0.5f * (hi[a] - lo[a]);
        s[a] = e > 0.f ? e : 1.0f;
    }
}

__device__ __forceinline__ unsigned short half_bits(_Float16 h) { return __builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ _Float16 half_from_bits(unsigned short b) { return __builtin_bit_cast(_Float16, b); }
// next representable half towards -inf / +inf (finite input)
__device__ __forceinline__ _Float16 half_next(_Float16 h, bool up)
{
    unsigned short b = half_bits(h);
    const bool neg = (b & 0x8000u) != 0;
    if ((b & 0x7fffu) == 0) return half_from_bits(up ? 0x0001u : 0x8001u);
    if (neg == up) b -= 1; else b += 1;      // moving towards zero shrinks the magnitude
    return half_from_bits(b);
}

__device__ __forceinline__ _Float16 tree_encode_half(float w, float c, float s, bool upper)
{
    const float u = (w - c) / s;
    _Float16 h = (_Float16)u;                                  // round to nearest; then walk outwards as far as needed
    if (!(u - u == 0.f)) return h;                             // +-inf (empty padding boxes) and NaN pass through
    for (int guard = 0; guard < 8; guard++) {
        const float d = __builtin_fmaf((float)h, s, c);
        if (upper ? d >= w : d <= w) break;
        h = half_next(h, upper);
    }
    return h;
}

__global__ __launch_bounds__(256) void tree_pack_pairs_half_kernel(const float4* __restrict__ boxes, int n_internal, uint4* __restrict__ out)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_internal) return;
    float c[3], s[3];
    tree_half_frame(boxes[0], boxes[1], c, s);
    const int l = 2 * p + 1;
    const float4 b4[4] = {boxes[2 * (size_t)l], boxes[2 * (size_t)l + 1], boxes[2 * (size_t)l + 2], boxes[2 * (size_t)l + 3]};
    unsigned short h[16] = {0};
    for (int q = 0; q < 4; q++) {                              // l.lo, l.hi, r.lo, r.hi
        const float w[3] = {b4[q].x, b4[q].y, b4[q].z};
        for (int a = 0; a < 3; a++) h[3 * q + a] = half_bits(tree_encode_half(w[a], c[a], s[a], (q & 1) != 0));
    }
    uint4 r0, r1;
    r0.x = h[0] | ((unsigned)h[1] << 16); r0.y = h[2] | ((unsigned)h[3] << 16); r0.z = h[4] | ((unsigned)h[5] << 16); r0.w = h[6] | ((unsigned)h[7] << 16);
    r1.x = h[8] | ((unsigned)h[9] << 16); r1.y = h[10] | ((unsigned)h[11] << 16); r1.z = 0; r1.w = 0;
    out[2 * (size_t)p] = r0;
    out[2 * (size_t)p + 1] = r1;
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
    if (a.n_pad > 1) {
        hipLaunchKernelGGL(tree_pack_pairs_kernel, dim3((a.n_pad - 1 + 255) / 256), dim3(256), 0, s, a.boxes, a.n_pad - 1, a.pairs);
        if (a.pairs_half)
            hipLaunchKernelGGL(tree_pack_pairs_half_kernel, dim3((a.n_pad - 1 + 255) / 256), dim3(256), 0, s, a.boxes, a.n_pad - 1, a.pairs_half);
    }
    hipLaunchKernelGGL(tree_pack_leaves_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, s, a.pts, n_slots, reinterpret_cast<float*>(a.leaf_soa),
                       a.leaf_idx);
    if (a.quads != nullptr && a.quad_levels > 0) {
        const unsigned int records = quad_level_offset(a.quad_levels);
        hipLaunchKernelGGL(tree_pack_quads_kernel, dim3((records + 255) / 256), dim3(256), 0, s, a.boxes, a.quad_levels, a.quad_parity, a.quads);
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

// lower bound of |q - s|^2 over q in [lo,hi], s in [glo,ghi], rounded like a distance (see the header comment)
template <bool FMA>
__device__ __forceinline__ float box_bound(const float4 lo, const float4 hi, const float glo[3], const float ghi[3])
{
    const float ex = fmaxf(fmaxf(lo.x - ghi[0], glo[0] - hi.x), 0.f);
    const float ey = fmaxf(fmaxf(lo.y - ghi[1], glo[1] - hi.y), 0.f);
    const float ez = fmaxf(fmaxf(lo.z - ghi[2], glo[2] - hi.z), 0.f);
    return sq3<FMA>(ex, ey, ez);
}

__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// Wave64 max/min returning a wave-uniform value: four DPP steps reduce each 16-lane row in the VALU (quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror, row_mirror), then the four row results are combined through v_readlane.  No LDS
// permute round trips (a __shfl_xor ladder is six ds_bpermute hops on the wave's critical path).
template <bool IS_MAX>
__device__ __forceinline__ float wave_reduce_f(float v)
{
    auto op = [](float a, float b) { return IS_MAX ? fmaxf(a, b) : fminf(a, b); };
#define MI_DPP_F(x, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), ctrl, 0xf, 0xf, false))
    v = op(v, MI_DPP_F(v, 0xB1));    // quad_perm:[1,0,3,2]
    v = op(v, MI_DPP_F(v, 0x4E));    // quad_perm:[2,3,0,1]
    v = op(v, MI_DPP_F(v, 0x141));   // row_half_mirror
    v = op(v, MI_DPP_F(v, 0x140));   // row_mirror
#undef MI_DPP_F
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return uniform_f(op(op(r0, r1), op(r2, r3)));
}
__device__ __forceinline__ float wave_min_f(float v) { return wave_reduce_f<false>(v); }
__device__ __forceinline__ float wave_max_f(float v) { return wave_reduce_f<true>(v); }

// v_writelane_b32: lane `lane` of `vec` <- val (both wave-uniform).  No clang builtin on this toolchain, hence asm: the lane
// select goes through M0 (written in the same statement), and the leading s_nop covers the "VALU-written SGPR used by
// v_writelane" wait states, which hipcc does not insert for operands of an asm statement.
__device__ __forceinline__ int write_lane(int vec, int val, int lane)
{
    const int sval = __builtin_amdgcn_readfirstlane(val);
    const int slane = __builtin_amdgcn_readfirstlane(lane);
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(vec) : "s"(sval), "s"(slane));
    return vec;
}

// One wave = 64*R consecutive (Morton-sorted) sources, lane l owns sources base + r*64 + l.
template <int R, bool FMA>
__global__ __launch_bounds__(256) void nn_tree_wave_kernel(const float4* __restrict__ tree_pts, const float4* __restrict__ tree_boxes,
                                                           int tree_n_pad, const float* __restrict__ sx, const float* __restrict__ sy,
                                                           const float* __restrict__ sz, int n, unsigned long long* __restrict__ keys,
                                                           const int* __restrict__ done_flag)
{
    if (done_flag != nullptr && *done_flag != 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
    const int base = wave * (64 * R);
    if (base >= n) return;

    float px[R], py[R], pz[R], best[R];
    unsigned int bidx[R];
    float glo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float ghi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = base + r * 64 + lane;
        const int ic = i < n ? i : n - 1;           // tail lanes shadow the last source (they never store)
        px[r] = sx[ic]; py[r] = sy[ic]; pz[r] = sz[ic];
        const unsigned long long k0 = keys[ic];      // starting candidate: the key already posted (KEY_INIT -> none)
        const unsigned int hi0 = (unsigned int)(k0 >> 32);
        best[r] = hi0 < 0x7f800000u ? __uint_as_float(hi0) : __builtin_inff();
        bidx[r] = hi0 < 0x7f800000u ? (unsigned int)(k0 & 0xffffffffull) : 0u;   // (inf, 0): nothing at +inf is ever accepted
        glo[0] = fminf(glo[0], px[r]); glo[1] = fminf(glo[1], py[r]); glo[2] = fminf(glo[2], pz[r]);
        ghi[0] = fmaxf(ghi[0], px[r]); ghi[1] = fmaxf(ghi[1], py[r]); ghi[2] = fmaxf(ghi[2], pz[r]);
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        glo[a] = uniform_f(wave_min_f(glo[a]));
        ghi[a] = uniform_f(wave_max_f(ghi[a]));
    }
    float gbest;   // the group's worst running best: a subtree farther than this cannot matter to any lane
    {
        float b = best[0];
#pragma unroll
        for (int r = 1; r < R; r++) b = fmaxf(b, best[r]);
        gbest = uniform_f(wave_max_f(b));
    }

    const int first_leaf = tree_n_pad - 1;
    // traversal stack across the lanes of two VGPRs (entry k lives in lane k), indexed by the scalar stack pointer
    int st_node = 0;
    int st_lb = 0;
    int sp = 0;
    int node = 0;
    bool have;
    {
        const float lb = uniform_f(box_bound<FMA>(tree_boxes[0], tree_boxes[1], glo, ghi));
        have = lb <= gbest && lb < __builtin_inff();
    }
    while (true) {
        if (!have) {
            bool found = false;
            while (sp > 0) {
                sp--;
                const float lb = __int_as_float(__builtin_amdgcn_readlane(st_lb, sp));
                if (lb <= gbest) { node = __builtin_amdgcn_readlane(st_node, sp); found = true; break; }
            }
            if (!found) break;
        }
        have = false;
        node = __builtin_amdgcn_readfirstlane(node);   // wave-uniform by construction; say so, so the loads below are scalar
        if (node >= first_leaf) {
            const float4* __restrict__ lp = tree_pts + (size_t)(node - first_leaf) * TREE_LEAF;   // wave-uniform: scalar loads
#pragma unroll
            for (int k = 0; k < TREE_LEAF; k++) {
                const float4 q = lp[k];
                const unsigned int j = (unsigned int)__float_as_int(q.w);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const float d = sq3<FMA>(q.x - px[r], q.y - py[r], q.z - pz[r]);
                    const bool better = (d < best[r]) | ((d == best[r]) & (j < bidx[r]));
                    best[r] = better ? d : best[r];
                    bidx[r] = better ? j : bidx[r];
                }
            }
            float b = best[0];
#pragma unroll
            for (int r = 1; r < R; r++) b = fmaxf(b, best[r]);
            gbest = uniform_f(wave_max_f(b));
        } else {
            const int l = 2 * node + 1;
            const float4* __restrict__ rec = tree_boxes + 2 * (size_t)l;   // {lo_l, hi_l, lo_r, hi_r}: one 64-byte scalar load
            const float lbl = uniform_f(box_bound<FMA>(rec[0], rec[1], glo, ghi));
            const float lbr = uniform_f(box_bound<FMA>(rec[2], rec[3], glo, ghi));
            const bool left_near = lbl <= lbr;
            const int near = left_near ? l : l + 1, far = left_near ? l + 1 : l;
            const float lbn = left_near ? lbl : lbr, lbf = left_near ? lbr : lbl;
            if (lbf <= gbest && lbf < __builtin_inff()) {
                st_node = write_lane(st_node, far, sp);
                st_lb = write_lane(st_lb, __float_as_int(lbf), sp);
                sp++;
            }
            if (lbn <= gbest && lbn < __builtin_inff()) { node = near; have = true; }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = base + r * 64 + lane;
        if (i < n && best[r] < __builtin_inff()) keys[i] = ((unsigned long long)__float_as_uint(best[r]) << 32) | bidx[r];
    }
}

// Per-lane form: every lane walks the hierarchy for its own source (degenerate group glo = ghi = s), per-lane stack in LDS
// (column per lane).  Divergent, but prunes with the lane's OWN best: the better form while the search radius is large
// compared with the extent of a wave's 64 neighbours (early ICP iterations).
template <bool FMA>
__global__ __launch_bounds__(256) void nn_tree_lane_kernel(const float4* __restrict__ tree_pts, const float4* __restrict__ tree_boxes,
                                                           int tree_n_pad, const float* __restrict__ sx, const float* __restrict__ sy,
                                                           const float* __restrict__ sz, int n, unsigned long long* __restrict__ keys,
                                                           const int* __restrict__ done_flag, int stack_depth)
{
    if (done_flag != nullptr && *done_flag != 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* st_node = reinterpret_cast<int*>(smem);
    float* st_lb = reinterpret_cast<float*>(smem + (size_t)stack_depth * 256 * sizeof(int));

    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float p[3] = {sx[i], sy[i], sz[i]};
    const unsigned long long k0 = keys[i];
    const unsigned int hi0 = (unsigned int)(k0 >> 32);
    float best = hi0 < 0x7f800000u ? __uint_as_float(hi0) : __builtin_inff();
    unsigned int bidx = hi0 < 0x7f800000u ? (unsigned int)(k0 & 0xffffffffull) : 0u;

    const int first_leaf = tree_n_pad - 1;
    int sp = 0;
    int node = 0;
    const float root_lb = box_bound<FMA>(tree_boxes[0], tree_boxes[1], p, p);
    bool have = root_lb <= best && root_lb < __builtin_inff();
    // next pending subtree that can still matter (deepest first), or have = false when the stack runs dry
    auto pop = [&]() {
        have = false;
        while (sp > 0) {
            sp--;
            const float lb = st_lb[sp * 256 + threadIdx.x];
            if (lb <= best) { node = st_node[sp * 256 + threadIdx.x]; have = true; break; }
        }
    };
    // "while-while" traversal: every lane first descends to its next leaf, THEN the wave scans leaves together.  With one
    // node-or-leaf step per trip the two bodies ran back to back under complementary masks on almost every trip.
    while (have) {
        while (have && node < first_leaf) {
            const int l = 2 * node + 1;
            const float4* __restrict__ rec = tree_boxes + 2 * (size_t)l;
            const float lbl = box_bound<FMA>(rec[0], rec[1], p, p);
            const float lbr = box_bound<FMA>(rec[2], rec[3], p, p);
            const bool left_near = lbl <= lbr;
            const int near = left_near ? l : l + 1, far = left_near ? l + 1 : l;
            const float lbn = left_near ? lbl : lbr, lbf = left_near ? lbr : lbl;
            if (lbf <= best && lbf < __builtin_inff()) {
                st_node[sp * 256 + threadIdx.x] = far;
                st_lb[sp * 256 + threadIdx.x] = lbf;
                sp++;
            }
            if (lbn <= best && lbn < __builtin_inff()) node = near;
            else pop();
        }
        if (!have) break;
        const float4* __restrict__ lp = tree_pts + (size_t)(node - first_leaf) * TREE_LEAF;
#pragma unroll
        for (int k = 0; k < TREE_LEAF; k++) {
            const float4 q = lp[k];
            const float d = sq3<FMA>(q.x - p[0], q.y - p[1], q.z - p[2]);
            const unsigned int j = (unsigned int)__float_as_int(q.w);
            const bool better = (d < best) | ((d == best) & (j < bidx));
            best = better ? d : best;
            bidx = better ? j : bidx;
        }
        pop();
    }
    if (best < __builtin_inff()) keys[i] = ((unsigned long long)__float_as_uint(best) << 32) | bidx;
}

// The same walk over the compact copies (NnTreeView::pairs / leaf_soa / leaf_idx): 3 instead of 4 loads per internal node and
// 3/4 of the loads per leaf (no index words).  The walk is bound by the number of divergent 16-byte loads the L1 has to serve
// (counters: ~12 line accesses per load instruction), so fewer loads per visit is what pays.  The winner is
// tracked by its sorted SLOT; the global index is fetched once at the end -- and on an exact tie, where the lower GLOBAL index
// must win (rare: duplicates, or the posted starting candidate met again).
template <bool FMA, bool HALF>
__global__ __launch_bounds__(256) void nn_tree_lane_compact_kernel(NnTreeView t, const float* __restrict__ sx, const float* __restrict__ sy,
                                                                   const float* __restrict__ sz, int n, unsigned long long* __restrict__ keys,
                                                                   const int* __restrict__ done_flag, int node_steps, int xcd_chunks, int leaf_steps)
{
    if (done_flag != nullptr && *done_flag != 0) return;
    // Pending subtrees: WHICH ones is a 32-bit trail in a register (bit l set = the sibling of this lane's level-l ancestor is
    // still to be visited; the heap numbering makes it computable), their BOUNDS sit in LDS, one word per level and lane.  Half
    // the LDS of a (node, bound) stack, and LDS is what caps the occupancy of this 32-VGPR kernel: 8 instead of 4 waves per SIMD
    // to hide the dependent loads behind.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* st_lb = reinterpret_cast<float*>(smem);
    const float4* __restrict__ pairs = t.pairs;
    const uint4* __restrict__ pairs_half = t.pairs_half;
    float hc[3] = {0.f, 0.f, 0.f}, hs[3] = {1.f, 1.f, 1.f};
    if (HALF) tree_half_frame(t.boxes[0], t.boxes[1], hc, hs);
    const float4* __restrict__ leaf_soa = t.leaf_soa;
    const int* __restrict__ leaf_idx = t.leaf_idx;

    // (Giving only 32 / 16 / 8 lanes of each wave a moving point -- shorter waves for a small moving cloud that leaves wave slots
    // empty anyway -- measured 0.31 / 0.33 / 0.45 ms against 0.33 ms at 125 000 points and slower everywhere above: a wave is as
    // long as its WORST lane's walk, not as the union of its lanes' walks.)
    // Which 256 moving points a block takes: workgroups are dealt to the 8 XCDs round-robin (block b runs on XCD b mod 8), so
    // block b takes chunk (b mod 8) * (grid / 8) + b / 8 -- every XCD walks ONE contiguous eighth of the Morton-sorted moving
    // cloud, in order, and its 4 MB L2 keeps that eighth of the hierarchy instead of a bit of everything.
    unsigned int chunk = blockIdx.x;
    if (xcd_chunks == 1) {
        const unsigned int per_xcd = gridDim.x >> 3;
        if (blockIdx.x < (per_xcd << 3)) chunk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    } else if (xcd_chunks > 1) {
        // runs of `xcd_chunks` consecutive chunks per XCD, the 8 XCDs taking neighbouring runs: an XCD still walks contiguous
        // stretches of the Morton curve (its L2 sees each leaf-level node once), but every XCD gets a share of every region
        const unsigned int run = (unsigned int)xcd_chunks;
        const unsigned int whole = gridDim.x / (8u * run) * (8u * run);
        if (blockIdx.x < whole) {
            const unsigned int x = blockIdx.x & 7u, j = blockIdx.x >> 3;
            chunk = ((j / run) * 8u + x) * run + j % run;
        }
    }
    const int i = (int)(chunk * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const float p[3] = {sx[i], sy[i], sz[i]};
    const unsigned long long k0 = keys[i];
    const unsigned int hi0 = (unsigned int)(k0 >> 32);
    float best = hi0 < 0x7f800000u ? __uint_as_float(hi0) : __builtin_inff();
    unsigned int bidx = hi0 < 0x7f800000u ? (unsigned int)(k0 & 0xffffffffull) : 0u;   // the winner's index while bslot < 0
    int bslot = -1;

    const int first_leaf = t.n_pad - 1;
    unsigned int trail = 0;
    int node = 0, level = 0;
    const float root_lb = box_bound<FMA>(t.boxes[0], t.boxes[1], p, p);
    bool have = root_lb <= best && root_lb < __builtin_inff();
    // next pending subtree that can still matter (deepest first), or have = false when none is left
    auto pop = [&]() {
        have = false;
        while (trail != 0) {
            const int b = 31 - __builtin_clz(trail);              // deepest pending level
            trail &= ~(1u << b);
            const int anc = ((node + 1) >> (level - b)) - 1;      // this lane's ancestor at level b ...
            node = ((anc + 1) ^ 1) - 1;                           // ... its sibling is the pending subtree
            level = b;                                            // (bits deeper than b are all clear now)
            if (st_lb[b * blockDim.x + threadIdx.x] <= best) { have = true; break; }
        }
    };
    // candidate at sorted slot `slot` with squared distance d: lexicographic (d, global index) minimum
    auto offer = [&](float d, int slot) {
        const bool tie = d == best;
        const bool lt = d < best;
        best = lt ? d : best;
        bslot = lt ? slot : bslot;
        if (tie) {
            const unsigned int j = (unsigned int)leaf_idx[slot];
            const unsigned int jb = bslot >= 0 ? (unsigned int)leaf_idx[bslot] : bidx;
            if (j < jb) bslot = slot;
        }
    };
    const int step_limit = node_steps > 0 ? node_steps : 0x7fffffff;
    while (have) {
        // at most `node_steps` node visits per round: lanes that are at their leaf do not wait for the wave's longest descent
        int step = 0;                                          // wave-uniform (kept in a scalar register)
        while (have && node < first_leaf) {
            step = __builtin_amdgcn_readfirstlane(step + 1);
            // (ending the node phase as soon as fewer than 4 ... 32 lanes are still descending measured 0.506 ... 0.558 against
            // 0.483 ms: the stragglers' visits are cheap next to an extra round)
            if (step > step_limit) break;
            float lbl, lbr;
            if (HALF) {
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                const uint4 r0 = pairs_half[2 * (size_t)node], r1 = pairs_half[2 * (size_t)node + 1];
                const h8 u = __builtin_bit_cast(h8, r0), w = __builtin_bit_cast(h8, r1);
                const float v[12] = {(float)u[0], (float)u[1], (float)u[2], (float)u[3], (float)u[4], (float)u[5],
                                     (float)u[6], (float)u[7], (float)w[0], (float)w[1], (float)w[2], (float)w[3]};
                float d[12];
#pragma unroll
                for (int q = 0; q < 12; q++) d[q] = __builtin_fmaf(v[q], hs[q % 3], hc[q % 3]);
                lbl = box_bound<FMA>(make_float4(d[0], d[1], d[2], 0.f), make_float4(d[3], d[4], d[5], 0.f), p, p);
                lbr = box_bound<FMA>(make_float4(d[6], d[7], d[8], 0.f), make_float4(d[9], d[10], d[11], 0.f), p, p);
            }
            const int l = 2 * node + 1;
            if (!HALF) {
                const float4* __restrict__ rec = pairs + 3 * (size_t)node;
                const float4 a = rec[0], b = rec[1], c = rec[2];
                lbl = box_bound<FMA>(make_float4(a.x, a.y, a.z, 0.f), make_float4(a.w, b.x, b.y, 0.f), p, p);
                lbr = box_bound<FMA>(make_float4(b.z, b.w, c.x, 0.f), make_float4(c.y, c.z, c.w, 0.f), p, p);
            }
            const bool left_near = lbl <= lbr;
            const float lbn = left_near ? lbl : lbr, lbf = left_near ? lbr : lbl;
            // step to the near child either way: the trail is relative to the current node, and the far child is "the sibling
            // of my ancestor at the new level"
            node = left_near ? l : l + 1;
            level += 1;
            if (lbf <= best && lbf < __builtin_inff()) {
                trail |= 1u << level;
                st_lb[level * blockDim.x + threadIdx.x] = lbf;
            }
            if (!(lbn <= best && lbn < __builtin_inff())) pop();
        }
        // up to `leaf_steps` leaf scans per round: a lane whose next pending subtree is a leaf again (the sibling leaf, mostly)
        // scans it right away instead of sitting through the next round's node visits
        for (int scan = 0; scan < leaf_steps; scan++) {
            if (!(have && node >= first_leaf)) break;
            const int leaf = node - first_leaf;
            const int slot0 = leaf * TREE_LEAF;
            const float4* __restrict__ lp = leaf_soa + (size_t)leaf * (3 * TREE_LEAF / 4);
#pragma unroll
            for (int c4 = 0; c4 < TREE_LEAF / 4; c4++) {
                const float4 X = lp[c4], Y = lp[TREE_LEAF / 4 + c4], Z = lp[2 * (TREE_LEAF / 4) + c4];
                offer(sq3<FMA>(X.x - p[0], Y.x - p[1], Z.x - p[2]), slot0 + 4 * c4);
                offer(sq3<FMA>(X.y - p[0], Y.y - p[1], Z.y - p[2]), slot0 + 4 * c4 + 1);
                offer(sq3<FMA>(X.z - p[0], Y.z - p[1], Z.z - p[2]), slot0 + 4 * c4 + 2);
                offer(sq3<FMA>(X.w - p[0], Y.w - p[1], Z.w - p[2]), slot0 + 4 * c4 + 3);
            }
            pop();
        }
    }
    if (best < __builtin_inff()) {
        const unsigned int j = bslot >= 0 ? (unsigned int)leaf_idx[bslot] : bidx;
        keys[i] = ((unsigned long long)__float_as_uint(best) << 32) | j;
    }
}

// The same walk with DYNAMIC work fetching.  A wave of the kernel above runs as long as its slowest lane while the others idle
// (counters: 1 875 load instructions per wave where a lane alone needs a few hundred).  Here a resident grid of waves pulls
// moving points from counters: whenever at least `refill_min` lanes of a wave have finished, those lanes store their
// results and take the next consecutive points (Morton neighbours, so the wave stays spatially coherent).  Refilling in
// batches, not lane by lane, keeps the new walks descending together while the rest wait at most one descent.  Per-point
// results do not depend on which lane or wave computes them: bit-identical to the static kernel.
// Measured at N = M = 1e6 (average search of the bench's 50 iterations / FETCH_SIZE per launch): static 0.814 ms / 57 MB,
// one counter 0.741 ms / 163 MB, one counter per XCD range (below) 0.750 ms / 27 MB; 9.20 -> 8.46 ms at 1e7.
template <bool FMA>
__global__ __launch_bounds__(256) void nn_tree_lane_dynamic_kernel(NnTreeView t, const float* __restrict__ sx, const float* __restrict__ sy,
                                                                   const float* __restrict__ sz, int n, unsigned long long* __restrict__ keys,
                                                                   const int* __restrict__ done_flag, unsigned int* __restrict__ next_point,
                                                                   int refill_min, int parts, int node_steps)
{
    if (done_flag != nullptr && *done_flag != 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* st_lb = reinterpret_cast<float*>(smem);
    const float4* __restrict__ pairs = t.pairs;
    const float4* __restrict__ leaf_soa = t.leaf_soa;
    const int* __restrict__ leaf_idx = t.leaf_idx;
    const float4 root_lo = t.boxes[0], root_hi = t.boxes[1];
    const int first_leaf = t.n_pad - 1;
    const int lane = threadIdx.x & 63;

    int i = -1;                      // the moving point this lane works on, -1 = none
    float p[3] = {0.f, 0.f, 0.f};
    float best = __builtin_inff();
    unsigned int bidx = 0u;
    int bslot = -1;
    unsigned int trail = 0;
    int node = 0, level = 0;
    bool have = false;
    bool exhausted = false;          // wave-uniform: the counter has run past n
    // first assignment static, like the kernel above (the four waves of a block share cache lines of 256 consecutive points);
    // the counter starts behind the resident grid's first points (set by the host)
    auto begin = [&](int q) {
        i = q;
        p[0] = sx[i]; p[1] = sy[i]; p[2] = sz[i];
        const unsigned long long k0 = keys[i];
        const unsigned int hi0 = (unsigned int)(k0 >> 32);
        best = hi0 < 0x7f800000u ? __uint_as_float(hi0) : __builtin_inff();
        bidx = hi0 < 0x7f800000u ? (unsigned int)(k0 & 0xffffffffull) : 0u;
        bslot = -1;
        trail = 0;
        node = 0;
        level = 0;
        const float root_lb = box_bound<FMA>(root_lo, root_hi, p, p);
        have = root_lb <= best && root_lb < __builtin_inff();
    };
    // The moving cloud is cut into `parts` contiguous ranges with a counter each (64 bytes apart).  parts = 8: one range per XCD
    // (workgroups are dealt to the XCDs round-robin, so block b runs on XCD b mod 8) -- an XCD then walks one eighth of the
    // Morton curve and its 4 MB L2 holds that eighth of the hierarchy instead of a bit of everything; a wave whose range is used
    // up takes from the next one, so the XCDs still balance each other.  parts = 1: one range, one counter.
    const int part0 = parts == 1 ? 0 : (int)(blockIdx.x % (unsigned int)parts);
    const int local_block = parts == 1 ? (int)blockIdx.x : (int)(blockIdx.x / (unsigned int)parts);
    const int static_blocks = (int)gridDim.x / parts;     // per range: blocks whose first 256 points are assigned statically
    auto range_lo = [&](int k) { return (int)((long long)n * k / parts); };
    if (local_block < static_blocks) {
        const int q = range_lo(part0) + local_block * 256 + (int)threadIdx.x;
        if (q < range_lo(part0 + 1)) begin(q);
    }
    int cur = part0, tried = 0;      // wave-uniform: the range this wave draws from, ranges found used up
    auto pop = [&]() {
        have = false;
        while (trail != 0) {
            const int b = 31 - __builtin_clz(trail);
            trail &= ~(1u << b);
            const int anc = ((node + 1) >> (level - b)) - 1;
            node = ((anc + 1) ^ 1) - 1;
            level = b;
            if (st_lb[b * 256 + threadIdx.x] <= best) { have = true; break; }
        }
    };
    auto offer = [&](float d, int slot) {
        const bool tie = d == best;
        const bool lt = d < best;
        best = lt ? d : best;
        bslot = lt ? slot : bslot;
        if (tie) {
            const unsigned int j = (unsigned int)leaf_idx[slot];
            const unsigned int jb = bslot >= 0 ? (unsigned int)leaf_idx[bslot] : bidx;
            if (j < jb) bslot = slot;
        }
    };
    for (;;) {
        // ---- finished lanes store their result; a batch of them takes the next points
        if (!have && i >= 0) {
            if (best < __builtin_inff()) {
                const unsigned int j = bslot >= 0 ? (unsigned int)leaf_idx[bslot] : bidx;
                keys[i] = ((unsigned long long)__float_as_uint(best) << 32) | j;
            }
            i = -1;
        }
        const unsigned long long idle = __ballot(!have);
        const int n_idle = __popcll(idle);
        if (n_idle == 64 && exhausted) break;
        if (!exhausted && n_idle >= refill_min) {
            // (a per-block stream of 256-point chunks through an LDS counter instead -- the four waves of a block keep sharing
            // cache lines, but blocks no longer balance each other -- measured 0.80 ms against 0.75 ms for this global counter)
            while (!exhausted) {
                const int lo = range_lo(cur), hi = range_lo(cur + 1);
                unsigned int taken = 0;          // points of this range handed out before this fetch
                if (lane == 0) taken = atomicAdd(next_point + 16 * cur, (unsigned int)n_idle);
                taken = (unsigned int)__builtin_amdgcn_readfirstlane((int)taken);
                if ((long long)lo + taken < (long long)hi) {
                    if (!have) {
                        const long long q = (long long)lo + taken + __popcll(idle & ((1ull << lane) - 1ull));
                        if (q < (long long)hi) begin((int)q);
                    }
                    break;
                }
                cur = cur + 1 == parts ? 0 : cur + 1;      // used up: the next range
                tried += 1;
                exhausted = tried >= parts;
            }
            if (__ballot(have) == 0ull) continue;      // nothing to walk (all pruned at the root, or no points left): store / refill again
        }
        // at most `node_steps` node visits per round (0 = until every lane is at a leaf): lanes that have reached their leaf
        // do not wait for the longest descent of the wave
        const int step_limit = node_steps > 0 ? node_steps : 0x7fffffff;
        int step = 0;                                          // wave-uniform (kept in a scalar register)
        while (have && node < first_leaf) {
            step = __builtin_amdgcn_readfirstlane(step + 1);
            if (step > step_limit) break;
            const float4* __restrict__ rec = pairs + 3 * (size_t)node;
            const float4 a = rec[0], b = rec[1], c = rec[2];
            const float lbl = box_bound<FMA>(make_float4(a.x, a.y, a.z, 0.f), make_float4(a.w, b.x, b.y, 0.f), p, p);
            const float lbr = box_bound<FMA>(make_float4(b.z, b.w, c.x, 0.f), make_float4(c.y, c.z, c.w, 0.f), p, p);
            const int l = 2 * node + 1;
            const bool left_near = lbl <= lbr;
            const float lbn = left_near ? lbl : lbr, lbf = left_near ? lbr : lbl;
            node = left_near ? l : l + 1;
            level += 1;
            if (lbf <= best && lbf < __builtin_inff()) {
                trail |= 1u << level;
                st_lb[level * 256 + threadIdx.x] = lbf;
            }
            if (!(lbn <= best && lbn < __builtin_inff())) pop();
        }
        if (have && node >= first_leaf) {
            const int leaf = node - first_leaf;
            const int slot0 = leaf * TREE_LEAF;
            const float4* __restrict__ lp = leaf_soa + (size_t)leaf * (3 * TREE_LEAF / 4);
#pragma unroll
            for (int c4 = 0; c4 < TREE_LEAF / 4; c4++) {
                const float4 X = lp[c4], Y = lp[TREE_LEAF / 4 + c4], Z = lp[2 * (TREE_LEAF / 4) + c4];
                offer(sq3<FMA>(X.x - p[0], Y.x - p[1], Z.x - p[2]), slot0 + 4 * c4);
                offer(sq3<FMA>(X.y - p[0], Y.y - p[1], Z.y - p[2]), slot0 + 4 * c4 + 1);
                offer(sq3<FMA>(X.z - p[0], Y.z - p[1], Z.z - p[2]), slot0 + 4 * c4 + 2);
                offer(sq3<FMA>(X.w - p[0], Y.w - p[1], Z.w - p[2]), slot0 + 4 * c4 + 3);
            }
            pop();
        }
    }
}

// The WIDE walk: the same hierarchy, the same exact pruning rule, two binary levels per visit (NnTreeView::quads).  One visit
// loads the boxes of four children (6 x 16 bytes), sorts them by bound, steps into the nearest and remembers the others that
// can still matter -- half the dependent steps of the binary walk, and a pending sibling costs no visit of its own.  Lanes of a
// wave are therefore out of step less often (counters: 15.6 of 64 lanes active per VALU instruction in the binary walk).
// Position bookkeeping: a lane is at (level, pos) = the pos-th wide node of its level; children are 4 pos + c, the ancestor at
// level k is pos >> 2 (level - k), the leaf at level quad_levels is leaf number pos.  Pending children: per level up to three
// words in LDS, farthest first, each the bound's bits with the child number in the two lowest mantissa bits (clearing them
// rounds a non-negative bound DOWN: the test `bound <= best` only prunes a hair less, never more); how many are pending per
// level is a 2-bit field of a register.  Same dynamic fetching, same per-XCD ranges as the kernel above.
template <bool FMA>
__global__ __launch_bounds__(256) void nn_tree_wide_kernel(NnTreeView t, const float* __restrict__ sx, const float* __restrict__ sy,
                                                           const float* __restrict__ sz, int n, unsigned long long* __restrict__ keys,
                                                           const int* __restrict__ done_flag, unsigned int* __restrict__ next_point,
                                                           int refill_min, int parts, int node_steps)
{
    if (done_flag != nullptr && *done_flag != 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned int* st = reinterpret_cast<unsigned int*>(smem);      // [level - 1][3][256], levels 1 .. quad_levels
    const float4* __restrict__ quads = t.quads;
    const float4* __restrict__ leaf_soa = t.leaf_soa;
    const int* __restrict__ leaf_idx = t.leaf_idx;
    const float4 root_lo = t.boxes[0], root_hi = t.boxes[1];
    const int leaf_level = t.quad_levels;
    const int lane = threadIdx.x & 63;

    int i = -1;
    float p[3] = {0.f, 0.f, 0.f};
    float best = __builtin_inff();
    unsigned int bidx = 0u;
    int bslot = -1;
    unsigned int pend = 0;           // 2 bits per level: pending children there
    unsigned int pos = 0;
    int level = 0;
    bool have = false;
    bool exhausted = false;
    auto begin = [&](int q) {
        i = q;
        p[0] = sx[i]; p[1] = sy[i]; p[2] = sz[i];
        const unsigned long long k0 = keys[i];
        const unsigned int hi0 = (unsigned int)(k0 >> 32);
        best = hi0 < 0x7f800000u ? __uint_as_float(hi0) : __builtin_inff();
        bidx = hi0 < 0x7f800000u ? (unsigned int)(k0 & 0xffffffffull) : 0u;
        bslot = -1;
        pend = 0;
        pos = 0;
        level = 0;
        const float root_lb = box_bound<FMA>(root_lo, root_hi, p, p);
        have = root_lb <= best && root_lb < __builtin_inff();
    };
    const int part0 = parts == 1 ? 0 : (int)(blockIdx.x % (unsigned int)parts);
    const int local_block = parts == 1 ? (int)blockIdx.x : (int)(blockIdx.x / (unsigned int)parts);
    const int static_blocks = (int)gridDim.x / parts;
    auto range_lo = [&](int k) { return (int)((long long)n * k / parts); };
    if (local_block < static_blocks) {
        const int q = range_lo(part0) + local_block * 256 + (int)threadIdx.x;
        if (q < range_lo(part0 + 1)) begin(q);
    }
    int cur = part0, tried = 0;
    // nearest pending child of the deepest level that has one; a level whose nearest is already too far is dropped whole
    auto pop = [&]() {
        have = false;
        while (pend != 0) {
            const int b = (31 - __builtin_clz(pend)) >> 1;
            const int count = (int)((pend >> (2 * b)) & 3u);
            const unsigned int w = st[(((b - 1) * 3) + (count - 1)) * 256 + threadIdx.x];
            if (__uint_as_float(w & ~3u) <= best) {
                pend -= 1u << (2 * b);
                pos = ((pos >> (2 * (level - b + 1))) << 2) | (w & 3u);
                level = b;
                have = true;
                break;
            }
            pend &= ~(3u << (2 * b));
        }
    };
    auto offer = [&](float d, int slot) {
        const bool tie = d == best;
        const bool lt = d < best;
        best = lt ? d : best;
        bslot = lt ? slot : bslot;
        if (tie) {
            const unsigned int j = (unsigned int)leaf_idx[slot];
            const unsigned int jb = bslot >= 0 ? (unsigned int)leaf_idx[bslot] : bidx;
            if (j < jb) bslot = slot;
        }
    };
    for (;;) {
        if (!have && i >= 0) {
            if (best < __builtin_inff()) {
                const unsigned int j = bslot >= 0 ? (unsigned int)leaf_idx[bslot] : bidx;
                keys[i] = ((unsigned long long)__float_as_uint(best) << 32) | j;
            }
            i = -1;
        }
        const unsigned long long idle = __ballot(!have);
        const int n_idle = __popcll(idle);
        if (n_idle == 64 && exhausted) break;
        if (!exhausted && n_idle >= refill_min) {
            while (!exhausted) {
                const int lo = range_lo(cur), hi = range_lo(cur + 1);
                unsigned int taken = 0;
                if (lane == 0) taken = atomicAdd(next_point + 16 * cur, (unsigned int)n_idle);
                taken = (unsigned int)__builtin_amdgcn_readfirstlane((int)taken);
                if ((long long)lo + taken < (long long)hi) {
                    if (!have) {
                        const long long q = (long long)lo + taken + __popcll(idle & ((1ull << lane) - 1ull));
                        if (q < (long long)hi) begin((int)q);
                    }
                    break;
                }
                cur = cur + 1 == parts ? 0 : cur + 1;
                tried += 1;
                exhausted = tried >= parts;
            }
            if (__ballot(have) == 0ull) continue;
        }
        const int step_limit = node_steps > 0 ? node_steps : 0x7fffffff;
        int step = 0;
        while (have && level < leaf_level) {
            step = __builtin_amdgcn_readfirstlane(step + 1);
            if (step > step_limit) break;
            const float4* __restrict__ rec = quads + 6 * (size_t)(quad_level_offset(level) + pos);
            const float4 LX = rec[0], LY = rec[1], LZ = rec[2], HX = rec[3], HY = rec[4], HZ = rec[5];
            float lb0 = box_bound<FMA>(make_float4(LX.x, LY.x, LZ.x, 0.f), make_float4(HX.x, HY.x, HZ.x, 0.f), p, p);
            float lb1 = box_bound<FMA>(make_float4(LX.y, LY.y, LZ.y, 0.f), make_float4(HX.y, HY.y, HZ.y, 0.f), p, p);
            float lb2 = box_bound<FMA>(make_float4(LX.z, LY.z, LZ.z, 0.f), make_float4(HX.z, HY.z, HZ.z, 0.f), p, p);
            float lb3 = box_bound<FMA>(make_float4(LX.w, LY.w, LZ.w, 0.f), make_float4(HX.w, HY.w, HZ.w, 0.f), p, p);
            unsigned int c0 = 0u, c1 = 1u, c2 = 2u, c3 = 3u;
            // sort the four (bound, child) pairs ascending: 5 compare-exchanges
#define MI_CX(A, B, CA, CB) { const bool sw = B < A; const float ta = sw ? B : A; const float tb = sw ? A : B; A = ta; B = tb; \
                              const unsigned int ua = sw ? CB : CA; const unsigned int ub = sw ? CA : CB; CA = ua; CB = ub; }
            MI_CX(lb0, lb1, c0, c1) MI_CX(lb2, lb3, c2, c3) MI_CX(lb0, lb2, c0, c2) MI_CX(lb1, lb3, c1, c3) MI_CX(lb1, lb2, c1, c2)
#undef MI_CX
            // children that can still matter are a prefix of the sorted four
            const int keep = (int)(lb0 <= best && lb0 < __builtin_inff()) + (int)(lb1 <= best && lb1 < __builtin_inff()) +
                             (int)(lb2 <= best && lb2 < __builtin_inff()) + (int)(lb3 <= best && lb3 < __builtin_inff());
            if (keep == 0) { pop(); continue; }
            level += 1;
            // pending at the new level, farthest first: slot 0 = the (keep-1)-th, ..., slot keep-2 = the second nearest
            unsigned int* slot = st + ((level - 1) * 3) * 256 + threadIdx.x;
            if (keep == 4) { slot[0] = (__float_as_uint(lb3) & ~3u) | c3; slot[256] = (__float_as_uint(lb2) & ~3u) | c2; slot[512] = (__float_as_uint(lb1) & ~3u) | c1; }
            else if (keep == 3) { slot[0] = (__float_as_uint(lb2) & ~3u) | c2; slot[256] = (__float_as_uint(lb1) & ~3u) | c1; }
            else if (keep == 2) { slot[0] = (__float_as_uint(lb1) & ~3u) | c1; }
            pend = (pend & ~(3u << (2 * level))) | ((unsigned int)(keep - 1) << (2 * level));
            pos = (pos << 2) | c0;
        }
        if (have && level >= leaf_level) {
            const int leaf = (int)pos;
            const int slot0 = leaf * TREE_LEAF;
            const float4* __restrict__ lp = leaf_soa + (size_t)leaf * (3 * TREE_LEAF / 4);
#pragma unroll
            for (int c4 = 0; c4 < TREE_LEAF / 4; c4++) {
                const float4 X = lp[c4], Y = lp[TREE_LEAF / 4 + c4], Z = lp[2 * (TREE_LEAF / 4) + c4];
                offer(sq3<FMA>(X.x - p[0], Y.x - p[1], Z.x - p[2]), slot0 + 4 * c4);
                offer(sq3<FMA>(X.y - p[0], Y.y - p[1], Z.y - p[2]), slot0 + 4 * c4 + 1);
                offer(sq3<FMA>(X.z - p[0], Y.z - p[1], Z.z - p[2]), slot0 + 4 * c4 + 2);
                offer(sq3<FMA>(X.w - p[0], Y.w - p[1], Z.w - p[2]), slot0 + 4 * c4 + 3);
            }
            pop();
        }
    }
}

// Per-lane form without any stack: the heap numbering makes ancestors and siblings computable, so a 32-bit "trail" (bit l set
// = the sibling of this lane's level-l ancestor is still to be visited) replaces the LDS stack.  Same visiting order as the
// stack form (deepest pending sibling first); a pending sibling's bound is re-computed from its box when it comes up
// (one 32-byte load) instead of being remembered.  No LDS at all, so occupancy is set by ~40 VGPRs alone.
template <bool FMA>
__global__ __launch_bounds__(256) void nn_tree_trail_kernel(const float4* __restrict__ tree_pts, const float4* __restrict__ tree_boxes,
                                                            int tree_n_pad, const float* __restrict__ sx, const float* __restrict__ sy,
                                                            const float* __restrict__ sz, int n, unsigned long long* __restrict__ keys,
                                                            const int* __restrict__ done_flag)
{
    if (done_flag != nullptr && *done_flag != 0) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float p[3] = {sx[i], sy[i], sz[i]};
    const unsigned long long k0 = keys[i];
    const unsigned int hi0 = (unsigned int)(k0 >> 32);
    float best = hi0 < 0x7f800000u ? __uint_as_float(hi0) : __builtin_inff();
    unsigned int bidx = hi0 < 0x7f800000u ? (unsigned int)(k0 & 0xffffffffull) : 0u;

    const int first_leaf = tree_n_pad - 1;
    unsigned int trail = 0;
    int node = 0, level = 0;
    const float root_lb = box_bound<FMA>(tree_boxes[0], tree_boxes[1], p, p);
    bool have = root_lb <= best && root_lb < __builtin_inff();
    while (true) {
        if (!have) {
            bool found = false;
            while (trail != 0) {
                const int b = 31 - __builtin_clz(trail);          // deepest pending level
                trail &= ~(1u << b);
                const int anc = ((node + 1) >> (level - b)) - 1;  // this lane's ancestor at level b ...
                const int sib = ((anc + 1) ^ 1) - 1;              // ... and its sibling
                const float lb = box_bound<FMA>(tree_boxes[2 * (size_t)sib], tree_boxes[2 * (size_t)sib + 1], p, p);
                node = sib; level = b;                            // (bits deeper than b are all clear now)
                if (lb <= best && lb < __builtin_inff()) { found = true; break; }
            }
            if (!found) break;
        }
        have = false;
        if (node >= first_leaf) {
            const float4* __restrict__ lp = tree_pts + (size_t)(node - first_leaf) * TREE_LEAF;
#pragma unroll
            for (int k = 0; k < TREE_LEAF; k++) {
                const float4 q = lp[k];
                const float d = sq3<FMA>(q.x - p[0], q.y - p[1], q.z - p[2]);
                const unsigned int j = (unsigned int)__float_as_int(q.w);
                const bool better = (d < best) | ((d == best) & (j < bidx));
                best = better ? d : best;
                bidx = better ? j : bidx;
            }
        } else {
            const int l = 2 * node + 1;
            const float4* __restrict__ rec = tree_boxes + 2 * (size_t)l;
            const float lbl = box_bound<FMA>(rec[0], rec[1], p, p);
            const float lbr = box_bound<FMA>(rec[2], rec[3], p, p);
            const bool left_near = lbl <= lbr;
            const float lbn = left_near ? lbl : lbr, lbf = left_near ? lbr : lbl;
            const bool go_near = lbn <= best && lbn < __builtin_inff();
            const bool keep_far = lbf <= best && lbf < __builtin_inff();
            // move to the near child either way: the trail is relative to the current node, and a pending far child is
            // "the sibling of my ancestor at level+1"
            node = left_near ? l : l + 1;
            level += 1;
            if (keep_far) trail |= 1u << level;
            have = go_near;
        }
    }
    if (best < __builtin_inff()) keys[i] = ((unsigned long long)__float_as_uint(best) << 32) | bidx;
}

// Dynamic work fetching for the default walk: MISLAM_TREE_DYNAMIC=1 forces it, =0 forbids it (read per call: the tests flip it);
// unset, a moving cloud that fits the resident grid has nothing to fetch and takes the static kernel.
static bool want_dynamic(int n, bool have_counter, int resident_blocks)
{
    if (!have_counter || resident_blocks <= 0) return false;
    const char* dyn_env = getenv("MISLAM_TREE_DYNAMIC");
    return dyn_env ? *dyn_env == '1' : (TREE_DYNAMIC_DEFAULT && (n + 255) / 256 > resident_blocks);
}

const char* nn_tree_kernel_name(int n, int R, bool have_counter, int resident_blocks)
{
    if (R < 0) return "nn_tree_trail_kernel";
    if (R > 0) return "nn_tree_wave_kernel";
    const char* compact_env = getenv("MISLAM_TREE_COMPACT");
    if (compact_env && *compact_env == '0') return "nn_tree_lane_kernel";
    const char* half_env = getenv("MISLAM_TREE_HALF");
    if (half_env && *half_env == '1') return "nn_tree_lane_compact_kernel";
    const char* wide_env = getenv("MISLAM_TREE_WIDE");
    if (have_counter && resident_blocks > 0 && (wide_env ? *wide_env == '1' : TREE_WIDE_DEFAULT)) return "nn_tree_wide_kernel";
    return want_dynamic(n, have_counter, resident_blocks) ? "nn_tree_lane_dynamic_kernel" : "nn_tree_lane_compact_kernel";
}

hipError_t nn_tree_query(const NnTreeView& t, const float* sx, const float* sy, const float* sz, int n, unsigned long long* keys,
                         const int* done_flag, int fma, int R, hipStream_t s, unsigned int* work_counter, int resident_blocks)
{
    if (n <= 0) return hipSuccess;
    const char* resident_env = getenv("MISLAM_TREE_RESIDENT");      // test hook: a tiny resident grid makes a small cloud refill often
    if (resident_env && atoi(resident_env) > 0) resident_blocks = atoi(resident_env);
    if (R == 0) {   // per-lane form with an LDS stack (default: 1.68 / 1.04 ms at N = M = 1e6 early / late; trail form 1.84 / 1.09)
        const int depth = t.height + 2;
        const size_t lds = (size_t)depth * 256 * 8;
        dim3 grid((n + 255) / 256), block(256);
        const char* compact_env = getenv("MISLAM_TREE_COMPACT");      // read per call: the tests flip these
        const bool compact = !(compact_env && *compact_env == '0');
        if (compact && t.pairs != nullptr) {   // same walk over the compact copies (default)
            const size_t lds_c = (size_t)(t.height + 1) * 256 * sizeof(float);      // one bound per level and lane
            // half-precision sibling records (2 loads instead of 3 per node; exact all the same, 62 parity tests): measured
            // SLOWER, 1.18 against 1.13 ms per search at N = M = 1e6 -- the 24 extra decode instructions per node cost more than
            // the 16 bytes save.  Kept as a tested alternative.
            const char* half_env = getenv("MISLAM_TREE_HALF");
            const bool half_nodes = half_env && *half_env == '1';
            const char* xcd_env = getenv("MISLAM_TREE_XCD_CHUNKS");        // static kernel: one contiguous eighth of the moving cloud per XCD
            const int xcd_chunks = xcd_env ? atoi(xcd_env) : TREE_XCD_CHUNKS;
            const char* leaf_steps_env = getenv("MISLAM_TREE_LEAF_STEPS");  // leaf scans per round
            const int leaf_steps = leaf_steps_env && atoi(leaf_steps_env) > 0 ? atoi(leaf_steps_env) : TREE_LEAF_STEPS;
            const char* steps_env = getenv("MISLAM_TREE_NODE_STEPS");      // node visits per round, 0 = no limit
            const int node_steps = steps_env ? atoi(steps_env) : TREE_NODE_STEPS;
            // the wide walk (MISLAM_TREE_WIDE=1 / =0; two binary levels per visit) always fetches dynamically
            const char* wide_env = getenv("MISLAM_TREE_WIDE");
            const bool wide = work_counter != nullptr && resident_blocks > 0 && t.quads != nullptr && t.quad_levels > 0 && !half_nodes &&
                              (wide_env ? *wide_env == '1' : TREE_WIDE_DEFAULT);
            if (wide) {
                const char* refill_env = getenv("MISLAM_TREE_REFILL");
                int refill_min = refill_env ? atoi(refill_env) : TREE_REFILL_MIN;
                refill_min = refill_min < 1 ? 1 : (refill_min > 64 ? 64 : refill_min);
                const size_t lds_w = (size_t)t.quad_levels * 3 * 256 * sizeof(unsigned int);
                const int per_cu = (int)std::min<size_t>(8, (160 * 1024) / lds_w);
                dim3 wgrid(std::min((n + 255) / 256, std::max(1, resident_blocks / 8 * per_cu)));
                const char* parts_env = getenv("MISLAM_TREE_PARTS");
                int parts = parts_env ? atoi(parts_env) : TREE_DYNAMIC_PARTS;
                if (parts < 1 || parts > TREE_DYNAMIC_PARTS || (int)wgrid.x < parts) parts = 1;
                hipError_t e = hipMemsetD32Async((hipDeviceptr_t)work_counter, (int)(wgrid.x / parts) * 256, 16 * TREE_DYNAMIC_PARTS, s);
                if (e != hipSuccess) return e;
                const int wide_steps = steps_env ? atoi(steps_env) : TREE_WIDE_NODE_STEPS;
                if (fma) hipLaunchKernelGGL(nn_tree_wide_kernel<true>, wgrid, block, lds_w, s, t, sx, sy, sz, n, keys, done_flag, work_counter, refill_min, parts, wide_steps);
                else hipLaunchKernelGGL(nn_tree_wide_kernel<false>, wgrid, block, lds_w, s, t, sx, sy, sz, n, keys, done_flag, work_counter, refill_min, parts, wide_steps);
                return hipGetLastError();
            }
            if (!half_nodes && want_dynamic(n, work_counter != nullptr, resident_blocks)) {
                const char* refill_env = getenv("MISLAM_TREE_REFILL");
                int refill_min = refill_env ? atoi(refill_env) : TREE_REFILL_MIN;
                refill_min = refill_min < 1 ? 1 : (refill_min > 64 ? 64 : refill_min);
                dim3 dgrid(std::min((n + 255) / 256, resident_blocks));
                const char* parts_env = getenv("MISLAM_TREE_PARTS");      // 8 = one range of the moving cloud per XCD (default), 1 = one range
                int parts = parts_env ? atoi(parts_env) : TREE_DYNAMIC_PARTS;
                if (parts < 1 || parts > TREE_DYNAMIC_PARTS || (int)dgrid.x < parts) parts = 1;
                // every range's counter starts behind the points its blocks take statically
                hipError_t e = hipMemsetD32Async((hipDeviceptr_t)work_counter, (int)(dgrid.x / parts) * 256, 16 * TREE_DYNAMIC_PARTS, s);
                if (e != hipSuccess) return e;
                if (fma) hipLaunchKernelGGL(nn_tree_lane_dynamic_kernel<true>, dgrid, block, lds_c, s, t, sx, sy, sz, n, keys, done_flag, work_counter, refill_min, parts, node_steps);
                else hipLaunchKernelGGL(nn_tree_lane_dynamic_kernel<false>, dgrid, block, lds_c, s, t, sx, sy, sz, n, keys, done_flag, work_counter, refill_min, parts, node_steps);
                return hipGetLastError();
            }
            if (half_nodes && t.pairs_half != nullptr) {
                if (fma) hipLaunchKernelGGL((nn_tree_lane_compact_kernel<true, true>), grid, block, lds_c, s, t, sx, sy, sz, n, keys, done_flag, node_steps, xcd_chunks, leaf_steps);
                else hipLaunchKernelGGL((nn_tree_lane_compact_kernel<false, true>), grid, block, lds_c, s, t, sx, sy, sz, n, keys, done_flag, node_steps, xcd_chunks, leaf_steps);
            } else {
                // (fetching both children's records with the node's own -- two levels per load trip -- measured slower at every
                // moving-cloud size, 0.35 against 0.33 ms at 125 000 and 1.15 against 0.81 ms at 1e6: most visits are short
                // excursions into pending subtrees, where the second record is wasted)
                // threads per block of the static kernel (MISLAM_TREE_BLOCK=64|128|256): a block's wave slots and LDS come free
                // only when its last wave is done, so smaller blocks refill the CU sooner.  Search ms at 1e6 / 1e7 / 1e5:
                //   256: 0.484 / 4.95 / 0.144    128: 0.478 / 4.91 / 0.137    64: 0.475 / 5.02 / 0.138
                const char* block_env = getenv("MISLAM_TREE_BLOCK");
                const int tb_req = block_env ? atoi(block_env) : TREE_BLOCK_THREADS;
                const int tb = tb_req == 64 || tb_req == 128 || tb_req == 256 ? tb_req : TREE_BLOCK_THREADS;
                const dim3 sgrid((n + tb - 1) / tb), sblock(tb);
                const size_t slds = (size_t)(t.height + 1) * tb * sizeof(float);
                const int sruns = xcd_chunks > 1 ? xcd_chunks * (256 / tb) : xcd_chunks;
                if (fma) hipLaunchKernelGGL((nn_tree_lane_compact_kernel<true, false>), sgrid, sblock, slds, s, t, sx, sy, sz, n, keys, done_flag, node_steps, sruns, leaf_steps);
                else hipLaunchKernelGGL((nn_tree_lane_compact_kernel<false, false>), sgrid, sblock, slds, s, t, sx, sy, sz, n, keys, done_flag, node_steps, sruns, leaf_steps);
            }
            return hipGetLastError();
        }
        if (fma) hipLaunchKernelGGL(nn_tree_lane_kernel<true>, grid, block, lds, s, t.pts, t.boxes, t.n_pad, sx, sy, sz, n, keys, done_flag, depth);
        else hipLaunchKernelGGL(nn_tree_lane_kernel<false>, grid, block, lds, s, t.pts, t.boxes, t.n_pad, sx, sy, sz, n, keys, done_flag, depth);
        return hipGetLastError();
    }
    if (R < 0) {    // per-lane form, stackless trail (MISLAM_TREE_R=-1): no LDS, but re-loads a box per pending sibling
        if (t.height > 30) return hipErrorInvalidValue;
        dim3 grid((n + 255) / 256), block(256);
        if (fma) hipLaunchKernelGGL(nn_tree_trail_kernel<true>, grid, block, 0, s, t.pts, t.boxes, t.n_pad, sx, sy, sz, n, keys, done_flag);
        else hipLaunchKernelGGL(nn_tree_trail_kernel<false>, grid, block, 0, s, t.pts, t.boxes, t.n_pad, sx, sy, sz, n, keys, done_flag);
        return hipGetLastError();
    }
    // R = 1, 2: wave-cooperative form
    if (t.height + 2 > 64) return hipErrorInvalidValue;   // the stack is one VGPR wide
    const int per_block = 4 * 64 * R;
    dim3 grid((n + per_block - 1) / per_block), block(256);
    if (R == 2) {
        if (fma) hipLaunchKernelGGL((nn_tree_wave_kernel<2, true>), grid, block, 0, s, t.pts, t.boxes, t.n_pad, sx, sy, sz, n, keys, done_flag);
        else hipLaunchKernelGGL((nn_tree_wave_kernel<2, false>), grid, block, 0, s, t.pts, t.boxes, t.n_pad, sx, sy, sz, n, keys, done_flag);
    } else {
        if (fma) hipLaunchKernelGGL((nn_tree_wave_kernel<1, true>), grid, block, 0, s, t.pts, t.boxes, t.n_pad, sx, sy, sz, n, keys, done_flag);
        else hipLaunchKernelGGL((nn_tree_wave_kernel<1, false>), grid, block, 0, s, t.pts, t.boxes, t.n_pad, sx, sy, sz, n, keys, done_flag);
    }
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
