// K7t -- the TRUNCATED exact E-step of the hybrid mode (ComputePMatrix(..., doTruncate = true, 1e-3f), coherentpointdrift.cpp:166 / :182-197),
// culled.  An affinity whose exponent lies below log(truncate) contributes EXACTLY zero to the denominator and to P1 / PX in the reference, and the
// hybrid mode only switches to this kernel once sigma^2 <= 0.015 sigma^2_0: the truncation radius sqrt(-2 sigma^2 ln 1e-3) = 3.72 sigma is then a
// few per cent of the cloud's extent and shrinks with every iteration -- yet round 4's truncated kernels (cpd_kernels.hip <.., TRUNC = true>)
// evaluated all N*M pairs and were 37 % SLOWER than the exact ones.
//
// Here both clouds are visited along a space-filling curve (the fixed cloud sorted once per registration; the moving cloud in the order of its
// ORIGINAL points' curve, also computed once -- a similarity transform keeps neighbours together), cut into tiles of 64 consecutive points with
// their bounding boxes (the moving tiles' boxes are taken anew for every E-step, with the gather of the current positions).  A workgroup of four
// waves owns a tile (lane = point); the OTHER cloud's tiles are dealt to the four waves round-robin, every lane tests ONE tile's box against the
// owner's (64 box tests per wave and step, no dependent chain), and a tile whose box is farther from the owner's than the truncation radius is
// skipped before any exp: every pair in it has its exponent below log(truncate) and would have added 0.0f.  A tile that survives is staged through
// LDS (one coalesced load, then broadcast reads) and evaluated for all 64 owners; the four waves' partial sums are added in wave order.
// What is tested and staged on the OTHER side are GROUPS of 16 points with boxes of their own, four groups to a load: 64 consecutive points of a
// surface along the curve span 1.0 x 1.5 x 1.1 length units on the bunny clouds (10 units across), and with whole tiles on both sides 11 tiles
// survived per owner tile even at a truncation radius of 0.12 (26 - 40 us per kernel with next to nothing in reach; a quarter of a tile is half as
// wide).
// (First form, measured on the bunny clouds: ONE wave per owner tile walking super-tiles and tiles through scalar loads -- a chain of ~250 dependent
// scalar round trips per wave, 0.10 ms per kernel with NOTHING in reach, slower than the every-pair kernel; profiles/r05_cpd_bench.log.)  Skipping a term that is exactly zero does not change a floating-point sum, so the only difference to the unculled kernel is the ORDER of
// the remaining terms (curve order instead of the caller's order): P1 / Pt1 / PX agree with it, and with the reference's sequential sums, to the
// same 2e-5 as before (tests/test_gpu_fgt.py, tests/test_gpu_cpd.py).
//
// The skip test is conservative: box gap^2 (an exact lower bound of every pair distance^2 up to two fp32 roundings) against
// r^2 (1 + 1e-4), r^2 = log(truncate) / (-0.5 / sigma^2) -- four orders of magnitude more slack than the roundings of either side.
//
// Outputs go where the unculled path puts them (Pt1, P1, PX in the CALLER's order, scattered through the sort orders), the contraction's operand
// xw4 is kept in curve order as well, and the M-step's x-sums / k-sums ride along (one row per workgroup, as the post kernels of cpd_kernels.hip
// leave them): the E-step is three launches -- gather + boxes, denominators, contraction.  Every sum has a fixed order (tiles ascending within a
// wave, waves in order): bitwise reproducible run to run.
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

// out[s] = in[order[s]] for the n points of a cloud (entries up to the next multiple of 64 replicate the last one: they keep a tile's tail in bounds
// and leave the boxes alone; the kernels below stage points past the cloud's end as points at infinity -- exponent -inf, affinity exactly 0), the box
// of every tile of 64 and of every GROUP of 16, component-major (box[q * count + index], q = lo xyz, hi xyz: the lane-parallel box tests read them
// coalesced).  One wave per tile.
__global__ __launch_bounds__(256) void cpd_trunc_gather_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                               const int* __restrict__ order, int n, float* __restrict__ ox, float* __restrict__ oy,
                                                               float* __restrict__ oz, float* __restrict__ tile_box, float* __restrict__ group_box,
                                                               const CpdState* __restrict__ state)
{
    if (state != nullptr && state->done != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    const int n_tiles = (n + CPD_TRUNC_TILE - 1) / CPD_TRUNC_TILE, n_groups = n_tiles * (CPD_TRUNC_TILE / CPD_TRUNC_GROUP);
    if (tile >= n_tiles) return;
    const int s = tile * CPD_TRUNC_TILE + lane;
    const int j = order[min(s, n - 1)];
    const float px = x[j], py = y[j], pz = z[j];
    ox[s] = px; oy[s] = py; oz[s] = pz;                      // (s < n_tiles * 64: the arrays are padded to whole tiles)
    float b[6] = {px, py, pz, px, py, pz};
#pragma unroll
    for (int m = 1; m < CPD_TRUNC_GROUP; m <<= 1) {          // within the lane's group of 16
#pragma unroll
        for (int q = 0; q < 3; q++) { b[q] = fminf(b[q], __shfl_xor(b[q], m, 64)); b[3 + q] = fmaxf(b[3 + q], __shfl_xor(b[3 + q], m, 64)); }
    }
    if ((lane & (CPD_TRUNC_GROUP - 1)) < 6) {
        const int q = lane & (CPD_TRUNC_GROUP - 1);
        group_box[(size_t)q * n_groups + tile * (CPD_TRUNC_TILE / CPD_TRUNC_GROUP) + lane / CPD_TRUNC_GROUP] = b[q];
    }
#pragma unroll
    for (int m = CPD_TRUNC_GROUP; m < 64; m <<= 1) {         // ... then across the tile's groups
#pragma unroll
        for (int q = 0; q < 3; q++) { b[q] = fminf(b[q], __shfl_xor(b[q], m, 64)); b[3 + q] = fmaxf(b[3 + q], __shfl_xor(b[3 + q], m, 64)); }
    }
    if (lane < 6) tile_box[(size_t)lane * n_tiles + tile] = b[lane];
}

// squared gap between box T of `boxes` (component-major, `count` per component) and the box `mine`: a lower bound of the squared distance of any
// two points, one from each
__device__ __forceinline__ float box_gap2(const float* __restrict__ boxes, int count, int T, const float (&mine)[6])
{
    const float lo_x = boxes[T], lo_y = boxes[(size_t)count + T], lo_z = boxes[2 * (size_t)count + T];
    const float hi_x = boxes[3 * (size_t)count + T], hi_y = boxes[4 * (size_t)count + T], hi_z = boxes[5 * (size_t)count + T];
    const float gx = fmaxf(fmaxf(lo_x - mine[3], mine[0] - hi_x), 0.f);
    const float gy = fmaxf(fmaxf(lo_y - mine[4], mine[1] - hi_y), 0.f);
    const float gz = fmaxf(fmaxf(lo_z - mine[5], mine[2] - hi_z), 0.f);
    return (gx * gx + gy * gy) + gz * gz;
}

// a wave's staging area belongs to that wave alone: what orders its LDS traffic is the wave's own program order (LDS operations of a wave complete
// in issue order) -- the compiler must not move them across these points, no workgroup barrier is needed
__device__ __forceinline__ void trunc_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

constexpr int CPD_TRUNC_WAVES = 4;
constexpr int CPD_TRUNC_PER_LOAD = CPD_TRUNC_TILE / CPD_TRUNC_GROUP;      // surviving groups one staging load brings in (a lane each point)

// The next (up to) four surviving groups of `todo` (bit b = group base + b * WAVES + wave), as this lane's point to stage -- lane L stages point
// L % 16 of the (L / 16)-th of them -- and how many groups that was.  Groups are taken in ascending order: every sum keeps a fixed order.
__device__ __forceinline__ int trunc_next_groups(unsigned long long& todo, int base, int wave, int lane, int& my_point)
{
    int ids[CPD_TRUNC_PER_LOAD], count = 0;
#pragma unroll
    for (int b = 0; b < CPD_TRUNC_PER_LOAD; b++) {
        ids[b] = 0;
        if (todo != 0ull) {
            ids[b] = base + (int)__builtin_ctzll(todo) * CPD_TRUNC_WAVES + wave;
            todo &= todo - 1ull;
            count++;
        }
    }
    const int slot = lane / CPD_TRUNC_GROUP;
    const int g = slot == 0 ? ids[0] : slot == 1 ? ids[1] : slot == 2 ? ids[2] : ids[3];
    my_point = slot < count ? g * CPD_TRUNC_GROUP + (lane & (CPD_TRUNC_GROUP - 1)) : -1;
    return count;
}

// ---- denominators: a workgroup per tile of the FIXED cloud (lane = fixed point x, in each of the four waves), the moving cloud's groups dealt to
// the waves
__global__ __launch_bounds__(64 * CPD_TRUNC_WAVES) void cpd_trunc_den_kernel(CpdTruncView v, double* __restrict__ xpartials)
{
    if (v.state->done != 0) return;
    __shared__ float4 stage[CPD_TRUNC_WAVES][CPD_TRUNC_TILE];
    __shared__ float part[CPD_TRUNC_WAVES][CPD_TRUNC_TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float c = v.state->constant;
    const float mult = -0.5f / v.state->sigma2;              // coherentpointdrift.cpp:176
    const float reach2 = (v.trunc_log / mult) * 1.0001f;     // pairs farther apart than this have index < trunc_log (slack: see the head of the file)
    const int n_tiles_a = (v.n + CPD_TRUNC_TILE - 1) / CPD_TRUNC_TILE;
    const int n_groups_y = (v.m + CPD_TRUNC_TILE - 1) / CPD_TRUNC_TILE * CPD_TRUNC_PER_LOAD;
    const float far = __builtin_inff();
    double acc[CPD_XSUMS] = {0};
    for (int tile = v.a_tile_lo + (int)blockIdx.x; tile < v.a_tile_hi; tile += gridDim.x) {      // (this rank's tiles: all of them on one GPU)
        const int i = tile * CPD_TRUNC_TILE + lane;
        const float ax = v.ax[i], ay = v.ay[i], az = v.az[i];                   // (padded to whole tiles)
        float mine[6];
#pragma unroll
        for (int q = 0; q < 6; q++) mine[q] = v.abox[(size_t)q * n_tiles_a + tile];   // wave-uniform -> scalar loads
        float sum = 0.f;
        for (int base = 0; base < n_groups_y; base += 64 * CPD_TRUNC_WAVES) {
            const int T = base + lane * CPD_TRUNC_WAVES + wave;                 // this lane's group to test: groups are dealt to the waves round-robin
            const bool in_reach = T < n_groups_y && box_gap2(v.ygroup, n_groups_y, min(T, n_groups_y - 1), mine) <= reach2;
            unsigned long long todo = __builtin_amdgcn_ballot_w64(in_reach);
            while (todo != 0ull) {                                              // (wave-uniform)
                int k;
                const int count = trunc_next_groups(todo, base, wave, lane, k);
                // (points past the cloud's end -- the last group's tail -- and the unused quarter-loads: at infinity, their affinity is exactly 0)
                const bool real = k >= 0 && k < v.m;
                const float4 pt = make_float4(real ? v.yx[k] : far, real ? v.yy[k] : far, real ? v.yz[k] : far, 0.f);
                trunc_wave_sync();                                              // (the last load's reads are done)
                stage[wave][lane] = pt;
                trunc_wave_sync();
                const int cnt = count * CPD_TRUNC_GROUP;
                for (int j = 0; j < cnt; j += 8) {                              // (cnt is a multiple of 16)
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const float4 y = stage[wave][j + u];                    // one address for the whole wave: a broadcast read
                        sum += trunc_affinity(mult * trunc_sq_dist(ax, ay, az, y.x, y.y, y.z), v.trunc_log);
                    }
                }
            }
        }
        part[wave][lane] = sum;
        __syncthreads();
        if (wave == 0 && i < v.n) {
            sum = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
            // denominator += constant; pt1(x) = 1 - constant / denominator   (:204-206) -- and the contraction's operand, as cpd_post_den_kernel
            const float den = sum + c;
            const float w = 1.0f / den;
            const float pt1 = 1.0f - c / den;
            const float4 rec = make_float4(ax * w, ay * w, az * w, w);
            const int io = v.a_order[i];
            v.pt1[io] = pt1;
            v.xw4[i] = rec;                 // curve order: the contraction's operand
            v.xw4_caller[io] = rec;         // caller's order: what the stand-alone x-sums (cpd_xsums_kernel: log den) and anything else downstream reads
            acc[0] += (double)logf(1.0f / w);                                   // error -= log(denominator), :215
            acc[1] += (double)ax * pt1; acc[2] += (double)ay * pt1; acc[3] += (double)az * pt1;
            acc[4] += (double)(ax * ax) * pt1 + (double)(ay * ay) * pt1 + (double)(az * az) * pt1;     // :257
        }
        __syncthreads();
    }
    if (wave != 0) return;
#pragma unroll
    for (int q = 0; q < CPD_XSUMS; q++) {
        const double tot = wave_sum(acc[q]);
        if (lane == 0) xpartials[(size_t)blockIdx.x * CPD_XSUMS + q] = tot;
    }
}

// ---- contraction: a workgroup per tile of the MOVING cloud (lane = moving point k), the fixed cloud's groups (static boxes) dealt to the waves
__global__ __launch_bounds__(64 * CPD_TRUNC_WAVES) void cpd_trunc_contract_kernel(CpdTruncView v, double* __restrict__ kpartials)
{
    if (v.state->done != 0) return;
    __shared__ float4 stage_a[CPD_TRUNC_WAVES][CPD_TRUNC_TILE], stage_w[CPD_TRUNC_WAVES][CPD_TRUNC_TILE];
    __shared__ float part[4][CPD_TRUNC_WAVES][CPD_TRUNC_TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float mult = -0.5f / v.state->sigma2;
    const float reach2 = (v.trunc_log / mult) * 1.0001f;
    const int n_tiles_y = (v.m + CPD_TRUNC_TILE - 1) / CPD_TRUNC_TILE;
    const int n_groups_a = (v.n + CPD_TRUNC_TILE - 1) / CPD_TRUNC_TILE * CPD_TRUNC_PER_LOAD;
    const float far = __builtin_inff();
    double acc[CPD_KSUMS] = {0};
    for (int tile = v.y_tile_lo + (int)blockIdx.x; tile < v.y_tile_hi; tile += gridDim.x) {      // (this rank's tiles of the moving cloud: all of them on one GPU)
        const int k = tile * CPD_TRUNC_TILE + lane;
        const float yx = v.yx[k], yy = v.yy[k], yz = v.yz[k];
        float mine[6];
#pragma unroll
        for (int q = 0; q < 6; q++) mine[q] = v.ybox[(size_t)q * n_tiles_y + tile];
        float p1 = 0.f, pxx = 0.f, pxy = 0.f, pxz = 0.f;
        for (int base = 0; base < n_groups_a; base += 64 * CPD_TRUNC_WAVES) {
            const int T = base + lane * CPD_TRUNC_WAVES + wave;
            const bool in_reach = T < n_groups_a && box_gap2(v.agroup, n_groups_a, min(T, n_groups_a - 1), mine) <= reach2;
            unsigned long long todo = __builtin_amdgcn_ballot_w64(in_reach);
            while (todo != 0ull) {
                int x;
                const int count = trunc_next_groups(todo, base, wave, lane, x);
                const bool real = x >= 0 && x < v.n;
                const float4 pa = make_float4(real ? v.ax[x] : far, real ? v.ay[x] : far, real ? v.az[x] : far, 0.f);
                const float4 pw = real ? v.xw4[x] : make_float4(0.f, 0.f, 0.f, 0.f);
                trunc_wave_sync();
                stage_a[wave][lane] = pa; stage_w[wave][lane] = pw;
                trunc_wave_sync();
                const int cnt = count * CPD_TRUNC_GROUP;
                for (int j = 0; j < cnt; j += 4) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const float4 a = stage_a[wave][j + u], w = stage_w[wave][j + u];
                        const float p = trunc_affinity(mult * trunc_sq_dist(a.x, a.y, a.z, yx, yy, yz), v.trunc_log);
                        p1 += p * w.w;           // p1(k) += p/den          coherentpointdrift.cpp:210-211
                        pxx += p * w.x;          // px.row(k) += x * p/den  :212
                        pxy += p * w.y;
                        pxz += p * w.z;
                    }
                }
            }
        }
        part[0][wave][lane] = p1; part[1][wave][lane] = pxx; part[2][wave][lane] = pxy; part[3][wave][lane] = pxz;
        __syncthreads();
        if (wave == 0 && k < v.m) {
            float tot[4];
#pragma unroll
            for (int q = 0; q < 4; q++) tot[q] = ((part[q][0][lane] + part[q][1][lane]) + part[q][2][lane]) + part[q][3][lane];
            const int ko = v.b_order[k];
            v.p1[ko] = tot[0];
            v.px[3 * (size_t)ko] = tot[1]; v.px[3 * (size_t)ko + 1] = tot[2]; v.px[3 * (size_t)ko + 2] = tot[3];
            const float b[3] = {v.bx[ko], v.by[ko], v.bz[ko]};
            acc[0] += (double)tot[0];
            for (int r = 0; r < 3; r++) {
                acc[1 + r] += (double)b[r] * tot[0];
                for (int cc = 0; cc < 3; cc++) acc[4 + 3 * r + cc] += (double)b[r] * tot[1 + cc];
                acc[13] += (double)(b[r] * b[r]) * tot[0];                                       // :259
            }
        }
        __syncthreads();
    }
    if (wave != 0) return;
#pragma unroll
    for (int q = 0; q < CPD_KSUMS; q++) {
        const double tot = wave_sum(acc[q]);
        if (lane == 0) kpartials[(size_t)blockIdx.x * CPD_KSUMS + q] = tot;
    }
}

hipError_t cpd_trunc_gather(const float* x, const float* y, const float* z, const int* order, int n, float* ox, float* oy, float* oz,
                            float* tile_box, float* group_box, const CpdState* state, hipStream_t s)
{
    const int n_tiles = (n + CPD_TRUNC_TILE - 1) / CPD_TRUNC_TILE;
    hipLaunchKernelGGL(cpd_trunc_gather_kernel, dim3((n_tiles + 3) / 4), dim3(256), 0, s, x, y, z, order, n, ox, oy, oz, tile_box, group_box, state);
    return hipGetLastError();
}

hipError_t cpd_trunc_denominators(const CpdTruncView& v, double* xpartials, int nblocks, hipStream_t s)
{
    hipLaunchKernelGGL(cpd_trunc_den_kernel, dim3(nblocks), dim3(64 * CPD_TRUNC_WAVES), 0, s, v, xpartials);
    return hipGetLastError();
}

hipError_t cpd_trunc_contract(const CpdTruncView& v, double* kpartials, int nblocks, hipStream_t s)
{
    hipLaunchKernelGGL(cpd_trunc_contract_kernel, dim3(nblocks), dim3(64 * CPD_TRUNC_WAVES), 0, s, v, kpartials);
    return hipGetLastError();
}

}  // namespace mislam
