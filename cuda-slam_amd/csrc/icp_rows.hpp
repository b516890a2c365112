// Per-chunk partial sums of an ICP iteration ("rows") and their fixed-order reductions.
//
// Every kernel that sums over the moving cloud -- the fused search (nn_grid.hip), the stand-alone moments kernel K2 and the
// transform + error kernel K4/K5 (icp_kernels.hip) -- cuts it into the same chunks of ICP_ROW_POINTS (64) consecutive points and
// writes one row of ICP_ROW doubles per chunk: 16 moments { count, sum b (3), sum a (3), sum a_r b_c (9) } and 2 error sums
// { sum |a - b'|^2, kept pairs }.  Within a chunk the terms are added in ONE fixed tree (lane butterfly, then the waves in
// order), so a row does not depend on which kernel produced it: the registration is bitwise the same whichever search strategy
// ran, and bitwise reproducible run to run (no float atomics anywhere).  Rows are then summed in index order by
// icp_rows_reduce (many workgroups -> a few rows) and by the solve kernel (a few rows -> the state).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace mislam {

constexpr int ICP_ROW_POINTS = 64;                        // moving points per row = workgroup (one wave) of the producing kernels
constexpr int ICP_ROW = ICP_MOMENTS + ICP_ERRSUMS;        // 18 doubles
constexpr int ICP_ROW_WAVES = ICP_ROW_POINTS / 64;
constexpr int ICP_MAX_REDUCED_ROWS = 64;                  // rows left after icp_rows_reduce

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) { return __shfl_xor(v, mask, 64); }

// Sum 16 per-lane doubles over a wave with a HALVING butterfly: at each step a lane hands over the half of its values it does
// not keep, so 16 values cost 8+4+2+1+1+1 = 17 exchanges instead of 16*6 = 96.  Afterwards lane L holds the wave total of value
// k(L) = 8*bit5 + 4*bit4 + 2*bit3 + bit2 of L.  a + b is commutative bit for bit, so both partners of an exchange compute the
// same sum: the tree is fixed.
__device__ __forceinline__ double wave_sum16(const double (&v)[16], int lane)
{
    double w8[8], w4[4], w2[2];
    const bool u5 = (lane & 32) != 0, u4 = (lane & 16) != 0, u3 = (lane & 8) != 0, u2 = (lane & 4) != 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const double keep = u5 ? v[j + 8] : v[j], send = u5 ? v[j] : v[j + 8];
        w8[j] = keep + shfl_xor_f64(send, 32);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const double keep = u4 ? w8[j + 4] : w8[j], send = u4 ? w8[j] : w8[j + 4];
        w4[j] = keep + shfl_xor_f64(send, 16);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const double keep = u3 ? w4[j + 2] : w4[j], send = u3 ? w4[j] : w4[j + 2];
        w2[j] = keep + shfl_xor_f64(send, 8);
    }
    const double keep = u2 ? w2[1] : w2[0], send = u2 ? w2[0] : w2[1];
    double x = keep + shfl_xor_f64(send, 4);
    x = x + shfl_xor_f64(x, 2);
    x = x + shfl_xor_f64(x, 1);
    return x;
}

// two per-lane doubles: lanes < 32 end up with the wave total of e0, lanes >= 32 with that of e1
__device__ __forceinline__ double wave_sum2(double e0, double e1, int lane)
{
    const bool u5 = (lane & 32) != 0;
    double x = (u5 ? e1 : e0) + shfl_xor_f64(u5 ? e0 : e1, 32);
#pragma unroll
    for (int m = 16; m > 0; m >>= 1) x = x + shfl_xor_f64(x, m);
    return x;
}

// Workgroup of ICP_ROW_POINTS threads: columns [0,16) of `row` <- sum of mom over the workgroup.  `lds` = ICP_ROW_WAVES * 16 doubles.
__device__ __forceinline__ void row_store_moments(const double (&mom)[16], double* __restrict__ row, double* lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double x = wave_sum16(mom, lane);
    const int col = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
    if constexpr (ICP_ROW_WAVES == 1) {                        // one wave: its totals ARE the row
        if ((lane & 3) == 0) row[col] = x;
        return;
    }
    if ((lane & 3) == 0) lds[wave * 16 + col] = x;
    __syncthreads();
    if (threadIdx.x < 16) {
        double tot = lds[threadIdx.x];
#pragma unroll
        for (int w = 1; w < ICP_ROW_WAVES; w++) tot += lds[w * 16 + threadIdx.x];
        row[threadIdx.x] = tot;
    }
    __syncthreads();
}

// columns [16,18) of `row` <- sum of (e0, e1) over the workgroup.  `lds` = ICP_ROW_WAVES * 2 doubles.
__device__ __forceinline__ void row_store_error(double e0, double e1, double* __restrict__ row, double* lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double x = wave_sum2(e0, e1, lane);
    if constexpr (ICP_ROW_WAVES == 1) {
        if ((lane & 31) == 0) row[ICP_MOMENTS + (lane >> 5)] = x;
        return;
    }
    if ((lane & 31) == 0) lds[wave * 2 + (lane >> 5)] = x;
    __syncthreads();
    if (threadIdx.x < 2) {
        double tot = lds[threadIdx.x];
#pragma unroll
        for (int w = 1; w < ICP_ROW_WAVES; w++) tot += lds[w * 2 + threadIdx.x];
        row[ICP_MOMENTS + threadIdx.x] = tot;
    }
    __syncthreads();
}

// One pair's contribution to the 16 moments: b = moving point (current position), a = its matched fixed point.
__device__ __forceinline__ void pair_moments(double (&m)[16], float bxf, float byf, float bzf, float axf, float ayf, float azf)
{
    const double bx = bxf, by = byf, bz = bzf, ax = axf, ay = ayf, az = azf;
    m[0] = 1.0;
    m[1] = bx; m[2] = by; m[3] = bz;
    m[4] = ax; m[5] = ay; m[6] = az;
    m[7] = ax * bx;  m[8] = ax * by;  m[9] = ax * bz;
    m[10] = ay * bx; m[11] = ay * by; m[12] = ay * bz;
    m[13] = az * bx; m[14] = az * by; m[15] = az * bz;
}

}  // namespace mislam
