// K7t -- the TRUNCATED exact E-step of the hybrid mode (ComputePMatrix(..., doTruncate = true, 1e-3f), coherentpointdrift.cpp:166 / :182-197),
// culled.  An affinity whose exponent lies below log(truncate) contributes EXACTLY zero to the denominator and to P1 / PX in the reference, and the
// hybrid mode only switches to this kernel once sigma^2 <= 0.015 sigma^2_0: the truncation radius sqrt(-2 sigma^2 ln 1e-3) = 3.72 sigma is then a
// few per cent of the cloud's extent and shrinks with every iteration -- yet round 4's truncated kernels (cpd_kernels.hip <.., TRUNC = true>)
// evaluated all N*M pairs and were 37 % SLOWER than the exact ones.
//
// Here both clouds are visited along a space-filling curve (the fixed cloud sorted once per registration; the moving cloud in the order of its
// ORIGINAL points' curve, also computed once -- a similarity transform keeps neighbours together), cut into tiles of 64 consecutive points with
// their bounding boxes (the moving tiles' boxes are taken anew for every E-step, with the gather of the current positions), 16 tiles to a
// super-tile.  One wave owns a tile (lane = point) and walks the OTHER cloud's super-tiles and tiles; a (super-)tile whose box is farther from the
// wave's own box than the truncation radius is skipped before any exp: every pair in it has its exponent below log(truncate) and would have added
// 0.0f.  Skipping a term that is exactly zero does not change a floating-point sum, so the only difference to the unculled kernel is the ORDER of
// the remaining terms (curve order instead of the caller's order): P1 / Pt1 / PX agree with it, and with the reference's sequential sums, to the
// same 2e-5 as before (tests/test_gpu_fgt.py, tests/test_gpu_cpd.py).
//
// The skip test is conservative: box gap^2 (an exact lower bound of every pair distance^2 up to two fp32 roundings) against
// r^2 (1 + 1e-4), r^2 = log(truncate) / (-0.5 / sigma^2) -- four orders of magnitude more slack than the roundings of either side.
//
// Outputs go where the unculled path puts them (Pt1, P1, PX in the CALLER's order, scattered through the sort orders), the contraction's operand
// xw4 stays in curve order, and the M-step's x-sums / k-sums ride along (one row per workgroup, as the post kernels of cpd_kernels.hip leave
// them): the E-step is three launches -- gather + boxes, denominators, contraction.
#include <hip/hip_runtime.h>
#include <math.h>

#include "cpd_kernels.h"
#include "reduce.hpp"

namespace mislam {

// (the arithmetic of cpd_kernels.hip's sq_dist / exp_neg / affinity<true>, restated here: the two files must agree to the bit)
__device__ __forceinline__ float trunc_sq_dist(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = ax - bx, dy = ay - by, dz = az - bz;   // cloudAfter[x] - cloudTransformed[k], coherentpointdrift.cpp:190
    return (dx * dx + dy * dy) + dz * dz;
}
__device__ __forceinline__ float trunc_affinity(float index, float trunc_log)
{
    const float L_hi = 1.44269502162933349609375f, L_lo = 1.925963033500011e-08f;
    const float h = index * L_hi;
    float r = __builtin_fmaf(index, L_hi, -h);
    r = __builtin_fmaf(index, L_lo, r);
    const float e = __builtin_amdgcn_exp2f(h);
    const float p = __builtin_fmaf(e * r, 0.693147182464599609375f, e);
    return index < trunc_log ? 0.f : p;                      // :193-196
}

__device__ __forceinline__ float wave_min_f32(float v)
{
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v)
{
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

// out[s] = in[order[s]] for the n points of a cloud (entries up to the next multiple of 64 replicate the last one: they are never summed, they
// only keep the scalar loads of a tile's tail in bounds), box of every tile of 64 and of every super-tile of 16 tiles.
// One workgroup = one super-tile = 16 waves.
__global__ __launch_bounds__(CPD_TRUNC_TILE * CPD_TRUNC_SUPER) void cpd_trunc_gather_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                                           const float* __restrict__ z, const int* __restrict__ order, int n,
                                                                                           float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz,
                                                                                           float* __restrict__ tile_box, float* __restrict__ super_box,
                                                                                           const CpdState* __restrict__ state)
{
    if (state != nullptr && state->done != 0) return;
    __shared__ float lds[CPD_TRUNC_SUPER][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * CPD_TRUNC_SUPER + wave;
    const int n_tiles = (n + CPD_TRUNC_TILE - 1) / CPD_TRUNC_TILE;
    const int s = tile * CPD_TRUNC_TILE + lane;
    float b[6] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    if (tile < n_tiles) {
        const int j = order[min(s, n - 1)];
        const float px = x[j], py = y[j], pz = z[j];
        ox[s] = px; oy[s] = py; oz[s] = pz;                  // (s < n_tiles * 64: the arrays are padded to whole tiles)
        b[0] = wave_min_f32(px); b[1] = wave_min_f32(py); b[2] = wave_min_f32(pz);
        b[3] = wave_max_f32(px); b[4] = wave_max_f32(py); b[5] = wave_max_f32(pz);
        if (lane < 6) tile_box[(size_t)tile * 6 + lane] = b[lane];
    }
    if (lane < 6) lds[wave][lane] = b[lane];
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = lds[0][threadIdx.x];
        for (int w = 1; w < CPD_TRUNC_SUPER; w++) v = threadIdx.x < 3 ? fminf(v, lds[w][threadIdx.x]) : fmaxf(v, lds[w][threadIdx.x]);
        super_box[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// squared gap between two boxes (lo xyz, hi xyz): a lower bound of the squared distance of any two points, one from each
__device__ __forceinline__ float box_gap2(const float* __restrict__ a, const float (&b)[6])
{
    const float gx = fmaxf(fmaxf(a[0] - b[3], b[0] - a[3]), 0.f);
    const float gy = fmaxf(fmaxf(a[1] - b[4], b[1] - a[4]), 0.f);
    const float gz = fmaxf(fmaxf(a[2] - b[5], b[2] - a[5]), 0.f);
    return (gx * gx + gy * gy) + gz * gz;
}

// ---- denominators: one wave per tile of the FIXED cloud (lane = fixed point x), the moving cloud's tiles streamed through scalar loads
__global__ __launch_bounds__(64) void cpd_trunc_den_kernel(CpdTruncView v, double* __restrict__ xpartials)
{
    if (v.state->done != 0) return;
    const int lane = threadIdx.x;
    const float c = v.state->constant;
    const float mult = -0.5f / v.state->sigma2;              // coherentpointdrift.cpp:176
    const float reach2 = (v.trunc_log / mult) * 1.0001f;     // pairs farther apart than this have index < trunc_log (slack: see the head of the file)
    const int n_tiles_a = (v.n + CPD_TRUNC_TILE - 1) / CPD_TRUNC_TILE, n_tiles_y = (v.m + CPD_TRUNC_TILE - 1) / CPD_TRUNC_TILE;
    const int n_super_y = (n_tiles_y + CPD_TRUNC_SUPER - 1) / CPD_TRUNC_SUPER;
    double acc[CPD_XSUMS] = {0};
    for (int tile = blockIdx.x; tile < n_tiles_a; tile += gridDim.x) {
        const int i = tile * CPD_TRUNC_TILE + lane;
        const bool live = i < v.n;
        const float ax = v.ax[i], ay = v.ay[i], az = v.az[i];                   // (padded to whole tiles)
        float mine[6];
#pragma unroll
        for (int k = 0; k < 6; k++) mine[k] = v.abox[(size_t)tile * 6 + k];     // wave-uniform -> scalar loads
        float sum = 0.f;
        for (int S = 0; S < n_super_y; S++) {
            if (box_gap2(v.ysuper + (size_t)S * 6, mine) > reach2) continue;
            const int t_end = min((S + 1) * CPD_TRUNC_SUPER, n_tiles_y);
            for (int T = S * CPD_TRUNC_SUPER; T < t_end; T++) {
                if (box_gap2(v.ybox + (size_t)T * 6, mine) > reach2) continue;
                const int k0 = T * CPD_TRUNC_TILE, k1 = min(k0 + CPD_TRUNC_TILE, v.m);
                int k = k0;
                for (; k + CPD_T <= k1; k += CPD_T) {
#pragma unroll
                    for (int u = 0; u < CPD_T; u++) {
                        const float yx = v.yx[k + u], yy = v.yy[k + u], yz = v.yz[k + u];    // wave-uniform -> scalar loads
                        sum += trunc_affinity(mult * trunc_sq_dist(ax, ay, az, yx, yy, yz), v.trunc_log);
                    }
                }
                for (; k < k1; k++) sum += trunc_affinity(mult * trunc_sq_dist(ax, ay, az, v.yx[k], v.yy[k], v.yz[k]), v.trunc_log);
            }
        }
        if (!live) continue;
        // denominator += constant; pt1(x) = 1 - constant / denominator   (:204-206) -- and the contraction's operand, as cpd_post_den_kernel
        const float den = sum + c;
        const float w = 1.0f / den;
        const float pt1 = 1.0f - c / den;
        v.pt1[v.a_order[i]] = pt1;
        v.xw4[i] = make_float4(ax * w, ay * w, az * w, w);
        acc[0] += (double)logf(1.0f / w);                                       // error -= log(denominator), :215
        acc[1] += (double)ax * pt1; acc[2] += (double)ay * pt1; acc[3] += (double)az * pt1;
        acc[4] += (double)(ax * ax) * pt1 + (double)(ay * ay) * pt1 + (double)(az * az) * pt1;     // :257
    }
#pragma unroll
    for (int k = 0; k < CPD_XSUMS; k++) {
        const double tot = wave_sum(acc[k]);
        if (lane == 0) xpartials[(size_t)blockIdx.x * CPD_XSUMS + k] = tot;
    }
}

// ---- contraction: one wave per tile of the MOVING cloud (lane = moving point k), the fixed cloud's tiles (static boxes) streamed
__global__ __launch_bounds__(64) void cpd_trunc_contract_kernel(CpdTruncView v, double* __restrict__ kpartials)
{
    if (v.state->done != 0) return;
    const int lane = threadIdx.x;
    const float mult = -0.5f / v.state->sigma2;
    const float reach2 = (v.trunc_log / mult) * 1.0001f;
    const int n_tiles_a = (v.n + CPD_TRUNC_TILE - 1) / CPD_TRUNC_TILE, n_tiles_y = (v.m + CPD_TRUNC_TILE - 1) / CPD_TRUNC_TILE;
    const int n_super_a = (n_tiles_a + CPD_TRUNC_SUPER - 1) / CPD_TRUNC_SUPER;
    double acc[CPD_KSUMS] = {0};
    for (int tile = blockIdx.x; tile < n_tiles_y; tile += gridDim.x) {
        const int k = tile * CPD_TRUNC_TILE + lane;
        const bool live = k < v.m;
        const float yx = v.yx[k], yy = v.yy[k], yz = v.yz[k];
        float mine[6];
#pragma unroll
        for (int q = 0; q < 6; q++) mine[q] = v.ybox[(size_t)tile * 6 + q];
        float p1 = 0.f, pxx = 0.f, pxy = 0.f, pxz = 0.f;
        for (int S = 0; S < n_super_a; S++) {
            if (box_gap2(v.asuper + (size_t)S * 6, mine) > reach2) continue;
            const int t_end = min((S + 1) * CPD_TRUNC_SUPER, n_tiles_a);
            for (int T = S * CPD_TRUNC_SUPER; T < t_end; T++) {
                if (box_gap2(v.abox + (size_t)T * 6, mine) > reach2) continue;
                const int x0 = T * CPD_TRUNC_TILE, x1 = min(x0 + CPD_TRUNC_TILE, v.n);
                int x = x0;
                for (; x + CPD_T <= x1; x += CPD_T) {
#pragma unroll
                    for (int u = 0; u < CPD_T; u++) {
                        const float ax = v.ax[x + u], ay = v.ay[x + u], az = v.az[x + u];    // wave-uniform -> scalar loads
                        const float4 w = v.xw4[x + u];
                        const float p = trunc_affinity(mult * trunc_sq_dist(ax, ay, az, yx, yy, yz), v.trunc_log);
                        p1 += p * w.w;           // p1(k) += p/den          coherentpointdrift.cpp:210-211
                        pxx += p * w.x;          // px.row(k) += x * p/den  :212
                        pxy += p * w.y;
                        pxz += p * w.z;
                    }
                }
                for (; x < x1; x++) {
                    const float4 w = v.xw4[x];
                    const float p = trunc_affinity(mult * trunc_sq_dist(v.ax[x], v.ay[x], v.az[x], yx, yy, yz), v.trunc_log);
                    p1 += p * w.w; pxx += p * w.x; pxy += p * w.y; pxz += p * w.z;
                }
            }
        }
        if (!live) continue;
        const int ko = v.b_order[k];
        v.p1[ko] = p1;
        v.px[3 * (size_t)ko] = pxx; v.px[3 * (size_t)ko + 1] = pxy; v.px[3 * (size_t)ko + 2] = pxz;
        const float b[3] = {v.bx[ko], v.by[ko], v.bz[ko]};
        const float px[3] = {pxx, pxy, pxz};
        acc[0] += (double)p1;
        for (int r = 0; r < 3; r++) {
            acc[1 + r] += (double)b[r] * p1;
            for (int cc = 0; cc < 3; cc++) acc[4 + 3 * r + cc] += (double)b[r] * px[cc];
            acc[13] += (double)(b[r] * b[r]) * p1;                                               // :259
        }
    }
#pragma unroll
    for (int q = 0; q < CPD_KSUMS; q++) {
        const double tot = wave_sum(acc[q]);
        if (lane == 0) kpartials[(size_t)blockIdx.x * CPD_KSUMS + q] = tot;
    }
}

hipError_t cpd_trunc_gather(const float* x, const float* y, const float* z, const int* order, int n, float* ox, float* oy, float* oz,
                            float* tile_box, float* super_box, const CpdState* state, hipStream_t s)
{
    const int n_tiles = (n + CPD_TRUNC_TILE - 1) / CPD_TRUNC_TILE;
    const int n_super = (n_tiles + CPD_TRUNC_SUPER - 1) / CPD_TRUNC_SUPER;
    hipLaunchKernelGGL(cpd_trunc_gather_kernel, dim3(n_super), dim3(CPD_TRUNC_TILE * CPD_TRUNC_SUPER), 0, s, x, y, z, order, n, ox, oy, oz, tile_box, super_box, state);
    return hipGetLastError();
}

hipError_t cpd_trunc_denominators(const CpdTruncView& v, double* xpartials, int nblocks, hipStream_t s)
{
    hipLaunchKernelGGL(cpd_trunc_den_kernel, dim3(nblocks), dim3(64), 0, s, v, xpartials);
    return hipGetLastError();
}

hipError_t cpd_trunc_contract(const CpdTruncView& v, double* kpartials, int nblocks, hipStream_t s)
{
    hipLaunchKernelGGL(cpd_trunc_contract_kernel, dim3(nblocks), dim3(64), 0, s, v, kpartials);
    return hipGetLastError();
}

}  // namespace mislam
