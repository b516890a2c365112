// Fast-Gauss-Transform E-step of rigid CPD on gfx950: the reference's approximation-type "full" / "hybrid"
// (source/common/fgt.cpp, source/common/cpdutils.cpp:19-77; both of its builds run this part on the CPU).
//
// The transform is  v(q) = sum_i w_i exp(-|q - s_i|^2 / sigma^2):  the sources s are grouped into K cells by greedy
// farthest-point clustering, each cell keeps the p-truncated Hermite/Taylor coefficients of its sources about the cell mean
// (pd = C(p+2,3) of them), and a query sums, over the cells within sqrt(e)*sigma, one polynomial of degree < p.
//
//   kcenter   ONE workgroup of 1024 lanes: the farthest-point sweep is a chain of K dependent arg-max steps, so it is
//             latency-bound; points, distances and labels live in registers (<= 16 per lane) and the arg-max travels with
//             the winner's coordinates through DPP-free shuffles + one LDS exchange, one barrier per step.  Same arithmetic,
//             same first-maximum tie rule and same strict-< relabel rule as the reference, so the labels are identical.
//   members   stable radix sort of (label, point id) -> per-cell member lists in ascending point order
//   centres   one lane per cell, sequential fp32 sum in member order: bit for bit the reference's cell means
//   model     one lane per (cell, monomial): sequential fp32 accumulation over the cell's members in the reference's order
//   predict   one lane per query, cells broadcast from SGPRs, the polynomial evaluated by a nested Horner scheme whose
//             coefficients stream through scalar loads in traversal order (no per-lane monomial table)
//
// Everything is deterministic (no float atomics).  Differences from cpu-slam come from expf (1 ulp) and the Horner
// evaluation order only.
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "cpd_fgt.h"

namespace mislam {

__device__ __forceinline__ float len2(float x, float y, float z) { return (x * x + y * y) + z * z; }   // point.h:49-51

// ---------------------------------------------------------------------------------------------------------------
// KCenter (fgt.cpp:152-193)
// ---------------------------------------------------------------------------------------------------------------
struct ArgMax {
    float v;
    int i;
    float x, y, z;
};

__device__ __forceinline__ bool beats(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__device__ __forceinline__ ArgMax wave_argmax(ArgMax a)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ArgMax o;
        o.v = __shfl_down(a.v, off, 64);
        o.i = __shfl_down(a.i, off, 64);
        o.x = __shfl_down(a.x, off, 64);
        o.y = __shfl_down(a.y, off, 64);
        o.z = __shfl_down(a.z, off, 64);
        if (beats(o.v, o.i, a.v, a.i)) a = o;
    }
    return a;
}

// PER > 0: every lane keeps its PER points (id = lane + r * 1024), their distances and labels in registers.
// PER == 0: any n; distances and labels stay in global memory (each lane only ever touches its own entries).
template <int PER>
__global__ __launch_bounds__(1024) void fgt_kcenter_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ z, int n, int K, float* __restrict__ dist,
                                                           int* __restrict__ indx)
{
    __shared__ ArgMax s_best[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NR = PER > 0 ? PER : 1;
    float px[NR], py[NR], pz[NR], pd[NR];
    int pc[NR];
    if (PER > 0) {
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int j = min(tid + r * 1024, n - 1);
            px[r] = x[j]; py[r] = y[j]; pz[r] = z[j];
            pd[r] = 0.f; pc[r] = 0;
        }
    }
    float cx = x[1], cy = y[1], cz = z[1];          // the first centre is point 1 (fgt.cpp:162)
    for (int step = 0; step < K; step++) {
        ArgMax best = {-1.f, 0x7fffffff, 0.f, 0.f, 0.f};
        if (PER > 0) {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int j = tid + r * 1024;
                if (j < n) {
                    const float d = len2(px[r] - cx, py[r] - cy, pz[r] - cz);
                    float cur = pd[r];
                    if (step == 0) cur = d;
                    else if (d < cur) { cur = d; pc[r] = step; }         // strict <: fgt.cpp:187
                    pd[r] = cur;
                    if (cur > best.v) best = {cur, j, px[r], py[r], pz[r]};   // ascending j per lane: the first maximum stays
                }
            }
        } else {
            for (int j = tid; j < n; j += 1024) {
                const float qx = x[j], qy = y[j], qz = z[j];
                const float d = len2(qx - cx, qy - cy, qz - cz);
                float cur;
                if (step == 0) { cur = d; dist[j] = d; indx[j] = 0; }
                else {
                    cur = dist[j];
                    if (d < cur) { cur = d; dist[j] = d; indx[j] = step; }
                }
                if (cur > best.v) best = {cur, j, qx, qy, qz};
            }
        }
        if (step == K - 1) break;
        // next centre = FIRST maximum of the distance array (std::max_element, fgt.cpp:179)
        best = wave_argmax(best);
        const int buf = step & 1;       // double-buffered: a wave can run at most one barrier ahead of the slowest reader
        if (lane == 0) s_best[buf][wave] = best;
        __syncthreads();
        ArgMax w = s_best[buf][0];
#pragma unroll
        for (int q = 1; q < 16; q++) {
            const ArgMax o = s_best[buf][q];
            if (beats(o.v, o.i, w.v, w.i)) w = o;
        }
        cx = w.x; cy = w.y; cz = w.z;
    }
    if (PER > 0) {
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int j = tid + r * 1024;
            if (j < n) indx[j] = pc[r];
        }
    }
    (void)dist;
}

__global__ __launch_bounds__(256) void fgt_iota_kernel(int* __restrict__ iota, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) iota[i] = i;
}

// off[k] = first sorted position whose label is >= k (k = 0..K)
__global__ __launch_bounds__(256) void fgt_offsets_kernel(const unsigned int* __restrict__ keys, int n, int K, int* __restrict__ off)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k > K) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < (unsigned int)k) lo = mid + 1; else hi = mid;
    }
    off[k] = lo;
}

// Cell means: sequential fp32 sums in ascending point order, then * (1.0f / count)   (fgt.cpp:195-210)
__global__ __launch_bounds__(64) void fgt_centers_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ z, const int* __restrict__ memb,
                                                         const int* __restrict__ off, int K, float* __restrict__ xc)
{
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= K) return;
    const int j0 = off[k], j1 = off[k + 1];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int j = j0; j < j1; j++) {
        const int i = memb[j];
        sx += x[i]; sy += y[i]; sz += z[i];
    }
    const float inv = 1.0f / (float)(j1 - j0);     // an empty cell gives 0 * inf = NaN, exactly as the reference
    xc[3 * k] = sx * inv; xc[3 * k + 1] = sy * inv; xc[3 * k + 2] = sz * inv;
}

// ---------------------------------------------------------------------------------------------------------------
// ComputeA_k (fgt.cpp:246-302): lane (k, t) accumulates its monomial over the members of cell k in the reference's order.
// The monomial of exponents (a, b, c) is built the way the recursion builds it: the seed exp(-|dx|^2) times z c times, then
// y b times, then x a times (prods[t] = val * prods[parent]).
// ---------------------------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(128) void fgt_model_kernel(FgtClusters c, const float4* __restrict__ w4, float inv_sigma, FgtTables t,
                                                        float* __restrict__ B)
{
    const int k = blockIdx.x;
    const int m = blockIdx.y * 128 + threadIdx.x;
    if (m >= t.pd) return;
    const unsigned int e = t.mono[m];
    const int ea = e & 0xff, eb = (e >> 8) & 0xff, ec = (e >> 16) & 0xff;
    const float cx = c.xc[3 * k], cy = c.xc[3 * k + 1], cz = c.xc[3 * k + 2];
    float acc[W];
#pragma unroll
    for (int w = 0; w < W; w++) acc[w] = 0.f;
    const int j0 = c.off[k], j1 = c.off[k + 1];
    for (int j = j0; j < j1; j++) {
        const int i = c.memb[j];
        const float dx = (c.x[i] - cx) * inv_sigma, dy = (c.y[i] - cy) * inv_sigma, dz = (c.z[i] - cz) * inv_sigma;
        float pr = expf(-len2(dx, dy, dz));
        for (int q = 0; q < ec; q++) pr = dz * pr;
        for (int q = 0; q < eb; q++) pr = dy * pr;
        for (int q = 0; q < ea; q++) pr = dx * pr;
        if (W == 1) acc[0] += pr;                                  // weights of ones (cpdutils.cpp:42)
        else {
            const float4 w = w4[i];
            acc[0] += w.x * pr; acc[1 % W] += w.y * pr; acc[2 % W] += w.z * pr; acc[3 % W] += w.w * pr;
        }
    }
    const float ck = t.ck[m];
    const int h = t.hpos[m];
#pragma unroll
    for (int w = 0; w < W; w++) B[((size_t)w * c.K + k) * t.pd + h] = acc[w] * ck;     // fgt.cpp:299-305
}

// ---------------------------------------------------------------------------------------------------------------
// ComputeFGTPredict (fgt.cpp:88-150).  Coefficients of cell k are stored in Horner traversal order:
//   for a = p-1..0, for b = p-1-a..0, for c = p-1-a-b..0   ->  sum_a x^a sum_b y^b sum_c z^c B[a,b,c]
// so the innermost loop is r = r*z + B[h++] with B[h] wave-uniform (scalar loads, batched when P is a compile-time constant).
// ---------------------------------------------------------------------------------------------------------------
// pa[w] = sum_a x^a sum_b y^b sum_c z^c Bk[w][h(a,b,c)]; P > 0: compile-time order, fully unrolled; P == 0: any order.
template <int W, int P>
struct Horner {
    static __device__ __forceinline__ void eval(int, float dx, float dy, float dz, const float* __restrict__ Bk, size_t wstride, float (&pa)[W])
    {
        int h = 0;
#pragma unroll
        for (int a = P - 1; a >= 0; a--) {
            float pb[W];
#pragma unroll
            for (int w = 0; w < W; w++) pb[w] = 0.f;
#pragma unroll
            for (int b = P - 1 - a; b >= 0; b--) {
                float pc[W];
#pragma unroll
                for (int w = 0; w < W; w++) pc[w] = 0.f;
#pragma unroll
                for (int cdeg = P - 1 - a - b; cdeg >= 0; cdeg--, h++) {
#pragma unroll
                    for (int w = 0; w < W; w++) pc[w] = __builtin_fmaf(pc[w], dz, Bk[w * wstride + h]);
                }
#pragma unroll
                for (int w = 0; w < W; w++) pb[w] = __builtin_fmaf(pb[w], dy, pc[w]);
            }
#pragma unroll
            for (int w = 0; w < W; w++) pa[w] = __builtin_fmaf(pa[w], dx, pb[w]);
        }
    }
};

template <int W>
struct Horner<W, 0> {
    static __device__ __forceinline__ void eval(int p, float dx, float dy, float dz, const float* __restrict__ Bk, size_t wstride, float (&pa)[W])
    {
        int h = 0;
        for (int a = p - 1; a >= 0; a--) {
            float pb[W];
            for (int w = 0; w < W; w++) pb[w] = 0.f;
            for (int b = p - 1 - a; b >= 0; b--) {
                float pc[W];
                for (int w = 0; w < W; w++) pc[w] = 0.f;
                for (int cdeg = p - 1 - a - b; cdeg >= 0; cdeg--, h++)
                    for (int w = 0; w < W; w++) pc[w] = __builtin_fmaf(pc[w], dz, Bk[w * wstride + h]);
                for (int w = 0; w < W; w++) pb[w] = __builtin_fmaf(pb[w], dy, pc[w]);
            }
            for (int w = 0; w < W; w++) pa[w] = __builtin_fmaf(pa[w], dx, pb[w]);
        }
    }
};

template <int W, int P>
__global__ __launch_bounds__(256) void fgt_predict_kernel(const float* __restrict__ qx, const float* __restrict__ qy,
                                                          const float* __restrict__ qz, int nq, const float* __restrict__ xc,
                                                          const float* __restrict__ B, int K, int pd, int p_runtime, float inv_sigma,
                                                          float e_param, float* __restrict__ v)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ic = min(i, nq - 1);
    const float x = qx[ic], y = qy[ic], z = qz[ic];
    float cell[W];
#pragma unroll
    for (int w = 0; w < W; w++) cell[w] = 0.f;
    const size_t wstride = (size_t)K * pd;
    for (int k = 0; k < K; k++) {
        const float dx = (x - xc[3 * k]) * inv_sigma, dy = (y - xc[3 * k + 1]) * inv_sigma, dz = (z - xc[3 * k + 2]) * inv_sigma;
        const float sum = len2(dx, dy, dz);
        if (sum > e_param) continue;                               // fgt.cpp:120 (a NaN cell is not skipped there either)
        const float e = expf(-sum);
        float pa[W];
#pragma unroll
        for (int w = 0; w < W; w++) pa[w] = 0.f;
        Horner<W, P>::eval(p_runtime, dx, dy, dz, B + (size_t)k * pd, wstride, pa);
#pragma unroll
        for (int w = 0; w < W; w++) cell[w] += e * pa[w];
    }
    if (i < nq) {
#pragma unroll
        for (int w = 0; w < W; w++) v[(size_t)w * nq + i] = cell[w];
    }
}

__global__ __launch_bounds__(256) void fgt_post_kt1_kernel(const float* __restrict__ kt1, const float* __restrict__ ax,
                                                           const float* __restrict__ ay, const float* __restrict__ az, int n, float ndi,
                                                           float* __restrict__ pt1, float4* __restrict__ xw4)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float inv = 1.0f / (kt1[i] + ndi);                       // cpdutils.cpp:49
    pt1[i] = 1.0f - ndi * inv;                                     // CalculatePt1, cpdutils.cpp:79-88
    xw4[i] = make_float4(ax[i] * inv, ay[i] * inv, az[i] * inv, inv);   // CalculateWeightsForPX :90-99; .w doubles as 1/denominator
}

__global__ __launch_bounds__(256) void fgt_post_px_kernel(const float* __restrict__ v, int m, float* __restrict__ p1, float* __restrict__ px)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= m) return;
    px[3 * (size_t)k] = v[k];
    px[3 * (size_t)k + 1] = v[(size_t)m + k];
    px[3 * (size_t)k + 2] = v[2 * (size_t)m + k];
    p1[k] = v[3 * (size_t)m + k];
}

// ---------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------
size_t fgt_sort_temp_bytes(int n)
{
    size_t bytes = 0;
    unsigned int* k = nullptr;
    int* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0u, (unsigned)FGT_KEY_BITS, (hipStream_t)0, false);
    return bytes;
}

hipError_t fgt_fill_iota(int* iota, int n, hipStream_t s)
{
    hipLaunchKernelGGL(fgt_iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, iota, n);
    return hipGetLastError();
}

hipError_t fgt_cluster(const FgtClusters& c, void* sort_temp, size_t sort_temp_bytes, hipStream_t s)
{
    if (c.n <= 4 * 1024) hipLaunchKernelGGL(fgt_kcenter_kernel<4>, dim3(1), dim3(1024), 0, s, c.x, c.y, c.z, c.n, c.K, c.dist, c.indx);
    else if (c.n <= 16 * 1024) hipLaunchKernelGGL(fgt_kcenter_kernel<16>, dim3(1), dim3(1024), 0, s, c.x, c.y, c.z, c.n, c.K, c.dist, c.indx);
    else hipLaunchKernelGGL(fgt_kcenter_kernel<0>, dim3(1), dim3(1024), 0, s, c.x, c.y, c.z, c.n, c.K, c.dist, c.indx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t temp = sort_temp_bytes;
    e = rocprim::radix_sort_pairs(sort_temp, temp, reinterpret_cast<const unsigned int*>(c.indx), c.keys_sorted, c.iota, c.memb,
                                  (size_t)c.n, 0u, (unsigned)FGT_KEY_BITS, s, false);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fgt_offsets_kernel, dim3((c.K + 1 + 255) / 256), dim3(256), 0, s, c.keys_sorted, c.n, c.K, c.off);
    hipLaunchKernelGGL(fgt_centers_kernel, dim3((c.K + 63) / 64), dim3(64), 0, s, c.x, c.y, c.z, c.memb, c.off, c.K, c.xc);
    return hipGetLastError();
}

hipError_t fgt_model(const FgtClusters& c, const float4* w4, float sigma, const FgtTables& t, float* B, hipStream_t s)
{
    const float inv = 1.0f / sigma;                                // fgt.cpp:260
    const dim3 grid(c.K, (t.pd + 127) / 128);
    if (w4) hipLaunchKernelGGL(fgt_model_kernel<4>, grid, dim3(128), 0, s, c, w4, inv, t, B);
    else hipLaunchKernelGGL(fgt_model_kernel<1>, grid, dim3(128), 0, s, c, w4, inv, t, B);
    return hipGetLastError();
}

hipError_t fgt_predict(const float* qx, const float* qy, const float* qz, int nq, const float* xc, const float* B, int K, int W,
                       float sigma, float e_param, const FgtTables& t, float* v, hipStream_t s)
{
    const float inv = 1.0f / sigma;                                // fgt.cpp:98
    const dim3 grid((nq + 255) / 256);
#define MISLAM_PREDICT(WW, PP) \
    hipLaunchKernelGGL((fgt_predict_kernel<WW, PP>), grid, dim3(256), 0, s, qx, qy, qz, nq, xc, B, K, t.pd, t.p, inv, e_param, v)
    if (t.p == 8) { if (W == 4) MISLAM_PREDICT(4, 8); else MISLAM_PREDICT(1, 8); }
    else { if (W == 4) MISLAM_PREDICT(4, 0); else MISLAM_PREDICT(1, 0); }
#undef MISLAM_PREDICT
    return hipGetLastError();
}

hipError_t fgt_post_kt1(const float* kt1, const float* ax, const float* ay, const float* az, int n, float ndi, float* pt1,
                        float4* xw4, hipStream_t s)
{
    hipLaunchKernelGGL(fgt_post_kt1_kernel, dim3((n + 255) / 256), dim3(256), 0, s, kt1, ax, ay, az, n, ndi, pt1, xw4);
    return hipGetLastError();
}

hipError_t fgt_post_px(const float* v, int m, float* p1, float* px, hipStream_t s)
{
    hipLaunchKernelGGL(fgt_post_px_kernel, dim3((m + 255) / 256), dim3(256), 0, s, v, m, p1, px);
    return hipGetLastError();
}

}  // namespace mislam
