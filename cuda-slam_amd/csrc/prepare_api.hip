// Input stage on the device: what Common::GetCloudsFromConfig (source/common/common.cpp:134-210) does to ONE cloud
// between LoadCloud and the registration call -- subcloud, normalisation to "cloud-spread", shuffle, noise, outliers, the known
// transformation -- with the reference's arithmetic, bit for bit.
//
// The reference's random decisions come from two host generators that differ between standard libraries (std::shuffle over a
// std::mt19937, rand()); they are not restated here: the caller draws them (as the host mirror in host/cloud_io.cpp does) and
// passes the OUTCOMES -- index vectors and unit draws -- exactly like the permutations of mi_nicp_register.
//
// Stages (all on the context's stream, no host round trip before the final download):
//   centre    sequential fp32 running sum of the subcloud in ITS order (std::accumulate, common.cpp:281-284): one wave per
//             component, 64 terms per step fed through v_readlane (the same primitive as MI_SUM_CPU_SEQUENTIAL)
//   extent    min / max of (p - centre): two-stage, order-independent, exact; scale = spread / largest span (common.cpp:57-95)
//   place     out[i] = ((raw[sub[shuffle[i]]] - centre) * scale) - (centre * -1)       gather + normalise + shuffle in one pass
//   noise     spread of the placed cloud (min / max again) -> reach = spread * intensity; flagged rows += u * (2 reach) - reach
//   outliers  bounds of the noised cloud -> appended rows = u * (hi - lo) + lo
//   move      p -> R p + t, glm's operation order (TransformPoint, common.cpp:45-49)
// Every stage is one or two passes over 12 bytes per point: HBM/latency-sized next to a registration; the sequential centre
// costs ~3.3 ms per million points (8 cycles per term) and is what makes the normalisation bit-exact.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "context.h"

using namespace mislam;

namespace {

constexpr int PREP_BLOCKS = 1024;

struct PrepState {
    float centre[3];
    float scale;
    int degenerate;       // |largest span| < 1e-15: NormalizeCloud returns the cloud unchanged (common.cpp:89-90)
    float lo[3], hi[3];   // result of the last bounds pass
};

__device__ __forceinline__ float seq_add64(float acc, float term)
{
#pragma unroll
    for (int j = 0; j < 64; j++) acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(term), j));
    return acc;
}

// centre of mass of raw[sub[i]], i = 0 .. n-1, in that order: 3 waves, wave w sums component w (common.cpp:281-284)
__global__ __launch_bounds__(192) void prep_centre_kernel(const float* __restrict__ raw, const int* __restrict__ sub, int n, PrepState* __restrict__ st)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        float term = 0.f;     // +0.0f leaves a running fp32 sum unchanged
        if (i < n) term = raw[3 * (size_t)(sub ? sub[i] : i) + wave];
        acc = seq_add64(acc, term);
    }
    if (lane == 0) st->centre[wave] = acc / (float)n;
}

// per-block min / max of (p - shift) over n points of `pts` (through `idx` when given); shift = st->centre or nothing
template <bool SHIFT>
__global__ __launch_bounds__(256) void prep_bounds_kernel(const float* __restrict__ pts, const int* __restrict__ idx, int n,
                                                          const PrepState* __restrict__ st, float* __restrict__ partials)
{
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    const float c[3] = {SHIFT ? st->centre[0] : 0.f, SHIFT ? st->centre[1] : 0.f, SHIFT ? st->centre[2] : 0.f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float* p = pts + 3 * (size_t)(idx ? idx[i] : i);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float v = SHIFT ? p[k] - c[k] : p[k];
            lo[k] = fminf(lo[k], v);
            hi[k] = fmaxf(hi[k], v);
        }
    }
    __shared__ float s[6][256];
#pragma unroll
    for (int k = 0; k < 3; k++) { s[k][threadIdx.x] = lo[k]; s[3 + k][threadIdx.x] = hi[k]; }
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                s[k][threadIdx.x] = fminf(s[k][threadIdx.x], s[k][threadIdx.x + w]);
                s[3 + k][threadIdx.x] = fmaxf(s[3 + k][threadIdx.x], s[3 + k][threadIdx.x + w]);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 6) partials[blockIdx.x * 6 + threadIdx.x] = s[threadIdx.x][0];
}

// finish a bounds pass; what = 0: keep lo/hi; what = 1: also derive the normalisation scale (common.cpp:86-92)
__global__ __launch_bounds__(256) void prep_bounds_finish_kernel(const float* __restrict__ partials, int nblocks, PrepState* __restrict__ st,
                                                                 int what, float size)
{
    __shared__ float s[6][256];
    float v[6];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = k < 3 ? __builtin_inff() : -__builtin_inff();
    for (int b = threadIdx.x; b < nblocks; b += 256) {
#pragma unroll
        for (int k = 0; k < 6; k++) v[k] = k < 3 ? fminf(v[k], partials[b * 6 + k]) : fmaxf(v[k], partials[b * 6 + k]);
    }
#pragma unroll
    for (int k = 0; k < 6; k++) s[k][threadIdx.x] = v[k];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
#pragma unroll
            for (int k = 0; k < 6; k++)
                s[k][threadIdx.x] = k < 3 ? fminf(s[k][threadIdx.x], s[k][threadIdx.x + w]) : fmaxf(s[k][threadIdx.x], s[k][threadIdx.x + w]);
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
#pragma unroll
    for (int k = 0; k < 3; k++) { st->lo[k] = s[k][0]; st->hi[k] = s[3 + k][0]; }
    if (what == 1) {
        const float span = fmaxf(fmaxf(s[3][0] - s[0][0], s[4][0] - s[1][0]), s[5][0] - s[2][0]);   // CalculateCloudSpread, common.cpp:72-79
        st->degenerate = fabsf(span) < 1e-15f ? 1 : 0;
        st->scale = size / span;
    }
}

// out[i] = normalised raw[sub[shuffle[i]]]   (NormalizeCloud's two GetAlignedCloud calls around the scaling, common.cpp:83-94)
__global__ __launch_bounds__(256) void prep_place_kernel(const float* __restrict__ raw, const int* __restrict__ sub, const int* __restrict__ shuffle,
                                                         int n, const PrepState* __restrict__ st, int normalise, float* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = shuffle ? shuffle[i] : i;
    const float* p = raw + 3 * (size_t)(sub ? sub[j] : j);
    const bool norm = normalise && st->degenerate == 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float v = p[k];
        if (norm) {
            const float c = st->centre[k];
            v = (v - c) * st->scale - c * -1.f;
        }
        out[3 * (size_t)i + k] = v;
    }
}

// AddNoiseToCloud (common.cpp:97-119): reach = spread * intensity from the bounds in st; row r += GetRandomPoint(-reach, reach)
// with GetRandomFloat(min, max) = u * (max - min) + min (testutils.cpp:7-11)
__global__ __launch_bounds__(256) void prep_noise_kernel(float* __restrict__ pts, const int* __restrict__ rows, const float* __restrict__ unit,
                                                         int n_noise, const PrepState* __restrict__ st, float intensity)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n_noise) return;
    const float spread = fmaxf(fmaxf(st->hi[0] - st->lo[0], st->hi[1] - st->lo[1]), st->hi[2] - st->lo[2]);
    const float reach = spread * intensity;
    const float mn = -reach, range = reach - mn;
    float* p = pts + 3 * (size_t)rows[q];
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = p[k] + (unit[3 * (size_t)q + k] * range + mn);
}

// AddOutliersToCloud (common.cpp:121-132): appended row q = GetRandomPoint(lo, hi) of the cloud's bounds
__global__ __launch_bounds__(256) void prep_outliers_kernel(float* __restrict__ pts, int n, const float* __restrict__ unit, int n_outliers,
                                                            const PrepState* __restrict__ st)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n_outliers) return;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float range = st->hi[k] - st->lo[k];
        pts[3 * (size_t)(n + q) + k] = unit[3 * (size_t)q + k] * range + st->lo[k];
    }
}

struct Rigid { float r[9]; float t[3]; };

// TransformPoint (common.cpp:45-49): glm's mat3 * vec3 -- m[0]*x + m[1]*y + m[2]*z, left to right -- then + t
__global__ __launch_bounds__(256) void prep_move_kernel(float* __restrict__ pts, int n, Rigid g)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float* p = pts + 3 * (size_t)i;
    const float x = p[0], y = p[1], z = p[2];
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = (g.r[k] * x + g.r[3 + k] * y + g.r[6 + k] * z) + g.t[k];
}

struct Scratch {     // per-call device scratch, freed on every exit path
    static constexpr int SLOTS = 12;      // mi_prepare_cloud takes 9
    void* p[SLOTS] = {nullptr};
    int used = 0;
    ~Scratch() { for (int i = 0; i < used; i++) (void)hipFree(p[i]); }
    template <typename T> int get(T** out, size_t count)
    {
        *out = nullptr;
        if (used >= SLOTS) { set_error("mi_prepare_cloud: scratch slots exhausted"); return MI_ERR_STATE; }
        MI_HIP(hipMalloc((void**)out, std::max<size_t>(count, 1) * sizeof(T)));
        p[used++] = *out;
        return MI_OK;
    }
};

template <typename T> int upload(mi_ctx* c, Scratch& s, const T* host, size_t count, T** dev)
{
    MI_TRY(s.get(dev, count));
    if (count) MI_HIP(hipMemcpyAsync(*dev, host, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return MI_OK;
}

int bounds_pass(mi_ctx* c, const float* pts, const int* idx, int n, bool shift, PrepState* st, float* partials, int what, float size)
{
    const int nb = std::max(1, std::min(PREP_BLOCKS, (n + 255) / 256));
    if (shift) hipLaunchKernelGGL(prep_bounds_kernel<true>, dim3(nb), dim3(256), 0, c->stream, pts, idx, n, st, partials);
    else hipLaunchKernelGGL(prep_bounds_kernel<false>, dim3(nb), dim3(256), 0, c->stream, pts, idx, n, st, partials);
    hipLaunchKernelGGL(prep_bounds_finish_kernel, dim3(1), dim3(256), 0, c->stream, partials, nb, st, what, size);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

}  // namespace

extern "C" void mi_prepare_params_default(mi_prepare_params* p)
{
    if (!p) return;
    *p = mi_prepare_params{};
    p->has_spread = 0;
    p->spread = 1.f;
    p->noise_intensity = 0.f;        // "noise-intensity-*"  configparser.cpp
    p->has_transform = 0;
    p->rotation[0] = p->rotation[4] = p->rotation[8] = 1.f;
}

extern "C" int mi_prepare_cloud(mi_ctx* c, const float* raw_xyz, int n_raw, const int* subcloud_idx, int subcloud_n, const int* shuffle_idx,
                                const int* noise_rows, const float* noise_unit, int n_noise, const float* outlier_unit, int n_outliers,
                                const mi_prepare_params* params, float* out_xyz, int* out_n)
{
    if (!c) { set_error("mi_prepare_cloud: null context"); return MI_ERR_INVALID_ARG; }
    if (!raw_xyz || !params || !out_xyz || !out_n) { set_error("mi_prepare_cloud: null argument"); return MI_ERR_INVALID_ARG; }
    if (n_raw < 1) { set_error("mi_prepare_cloud: empty cloud"); return MI_ERR_INVALID_ARG; }
    const int n = subcloud_idx ? subcloud_n : n_raw;
    if (n < 1 || n > n_raw) { set_error("mi_prepare_cloud: subcloud of %d out of %d points", n, n_raw); return MI_ERR_INVALID_ARG; }
    if (n_noise < 0 || n_noise > n || n_outliers < 0) { set_error("mi_prepare_cloud: n_noise %d, n_outliers %d", n_noise, n_outliers); return MI_ERR_INVALID_ARG; }
    if ((n_noise > 0 && (!noise_rows || !noise_unit)) || (n_outliers > 0 && !outlier_unit)) { set_error("mi_prepare_cloud: missing draws"); return MI_ERR_INVALID_ARG; }
    if ((long long)n + n_outliers > 0x7fffffffLL / 3) { set_error("mi_prepare_cloud: too many points"); return MI_ERR_INVALID_ARG; }
    // a bad index would be an out-of-bounds gather on the device: check on the host (O(n), like the upload itself)
    if (subcloud_idx)
        for (int i = 0; i < n; i++)
            if (subcloud_idx[i] < 0 || subcloud_idx[i] >= n_raw) { set_error("mi_prepare_cloud: subcloud_idx[%d] = %d outside [0, %d)", i, subcloud_idx[i], n_raw); return MI_ERR_INVALID_ARG; }
    if (shuffle_idx)
        for (int i = 0; i < n; i++)
            if (shuffle_idx[i] < 0 || shuffle_idx[i] >= n) { set_error("mi_prepare_cloud: shuffle_idx[%d] = %d outside [0, %d)", i, shuffle_idx[i], n); return MI_ERR_INVALID_ARG; }
    for (int q = 0; q < n_noise; q++)
        if (noise_rows[q] < 0 || noise_rows[q] >= n || (q > 0 && noise_rows[q] <= noise_rows[q - 1])) {
            set_error("mi_prepare_cloud: noise_rows must be ascending rows of the prepared cloud (entry %d = %d)", q, noise_rows[q]);
            return MI_ERR_INVALID_ARG;
        }
    MI_ENTER(c);

    Scratch s;
    float *d_raw, *d_out, *d_partials, *d_noise_unit = nullptr, *d_outlier_unit = nullptr;
    int *d_sub = nullptr, *d_shuffle = nullptr, *d_rows = nullptr;
    PrepState* d_st;
    const int n_total = n + n_outliers;
    MI_TRY(upload(c, s, raw_xyz, (size_t)3 * n_raw, &d_raw));
    if (subcloud_idx) MI_TRY(upload(c, s, subcloud_idx, (size_t)n, &d_sub));
    if (shuffle_idx) MI_TRY(upload(c, s, shuffle_idx, (size_t)n, &d_shuffle));
    MI_TRY(s.get(&d_out, (size_t)3 * n_total));
    MI_TRY(s.get(&d_partials, (size_t)6 * PREP_BLOCKS));
    MI_TRY(s.get(&d_st, 1));
    MI_HIP(hipMemsetAsync(d_st, 0, sizeof(PrepState), c->stream));

    if (params->has_spread) {
        hipLaunchKernelGGL(prep_centre_kernel, dim3(1), dim3(192), 0, c->stream, d_raw, d_sub, n, d_st);
        MI_HIP(hipGetLastError());
        MI_TRY(bounds_pass(c, d_raw, d_sub, n, true, d_st, d_partials, 1, params->spread));
    }
    hipLaunchKernelGGL(prep_place_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_raw, d_sub, d_shuffle, n, d_st, params->has_spread, d_out);
    MI_HIP(hipGetLastError());
    if (n_noise > 0) {
        MI_TRY(upload(c, s, noise_rows, (size_t)n_noise, &d_rows));
        MI_TRY(upload(c, s, noise_unit, (size_t)3 * n_noise, &d_noise_unit));
        MI_TRY(bounds_pass(c, d_out, nullptr, n, false, d_st, d_partials, 0, 0.f));
        hipLaunchKernelGGL(prep_noise_kernel, dim3((n_noise + 255) / 256), dim3(256), 0, c->stream, d_out, d_rows, d_noise_unit, n_noise, d_st,
                           params->noise_intensity);
        MI_HIP(hipGetLastError());
    }
    if (n_outliers > 0) {
        MI_TRY(upload(c, s, outlier_unit, (size_t)3 * n_outliers, &d_outlier_unit));
        MI_TRY(bounds_pass(c, d_out, nullptr, n, false, d_st, d_partials, 0, 0.f));
        hipLaunchKernelGGL(prep_outliers_kernel, dim3((n_outliers + 255) / 256), dim3(256), 0, c->stream, d_out, n, d_outlier_unit, n_outliers, d_st);
        MI_HIP(hipGetLastError());
    }
    if (params->has_transform) {
        Rigid g;
        for (int k = 0; k < 9; k++) g.r[k] = params->rotation[k];
        for (int k = 0; k < 3; k++) g.t[k] = params->translation[k];
        hipLaunchKernelGGL(prep_move_kernel, dim3((n_total + 255) / 256), dim3(256), 0, c->stream, d_out, n_total, g);
        MI_HIP(hipGetLastError());
    }
    MI_HIP(hipMemcpyAsync(out_xyz, d_out, (size_t)3 * n_total * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipStreamSynchronize(c->stream));
    *out_n = n_total;
    return MI_OK;
}

// loads this translation unit's code object at mi_ctx_create (kernels.h)
namespace mislam {
__global__ void preload_prepare_api_kernel() {}
hipError_t preload_prepare_api()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(preload_prepare_api_kernel));
}
}  // namespace mislam
