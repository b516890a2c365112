// Deterministic two-stage reductions shared by the ICP and CPD kernels: per-thread fp64 accumulators -> wave shuffle ->
// LDS -> one partial row per block; a single workgroup then sums the rows in a fixed order.  No float atomics, so every
// result is bitwise reproducible run to run.
#pragma once
#include <hip/hip_runtime.h>

namespace mislam {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum W doubles per thread over a 256-thread block; the result lands in partial_out[0..W) (written by thread 0..W-1).
template <int W>
__device__ __forceinline__ void block_sum_store(const double (&acc)[W], double* __restrict__ partial_out)
{
    __shared__ double lds[4][W];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < W; k++) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) lds[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < W) {
        const int k = threadIdx.x;
        partial_out[k] = ((lds[0][k] + lds[1][k]) + lds[2][k]) + lds[3][k];
    }
}

// One workgroup sums `nblocks` partial rows of W doubles in a fixed order into out[W] (LDS, valid after the barrier).
template <int W>
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partials, int nblocks, double (&out)[W], double* lds /*[256]*/)
{
    static_assert(256 % W == 0, "W must divide 256");
    constexpr int G = 256 / W;   // row groups
    const int k = threadIdx.x % W, g = threadIdx.x / W;
    double s = 0.0;
    // the adds stay in row order (fixed summation order); the unroll only keeps 8 independent loads in flight per lane --
    // one load per trip made this loop a chain of L2 round trips (25 us for 1024 rows)
#pragma unroll 8
    for (int b = g; b < nblocks; b += G) s += partials[(size_t)b * W + k];
    lds[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < W) {
        double tot = 0.0;
        for (int gg = 0; gg < G; gg++) tot += lds[gg * W + threadIdx.x];
        lds[threadIdx.x] = tot;   // safe: thread k only overwrites slot k, which only thread k reads (gg = 0)
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < W; i++) out[i] = lds[i];
    __syncthreads();
}


// acc + t_0 + t_1 + ... + t_63 added ONE BY ONE in lane order, the terms read out of the lanes' registers (v_readlane): how
// the reference's sequential fp32 running sums (std::accumulate, `sum +=` loops) are retraced bit for bit, 64 terms per step.
__device__ __forceinline__ float seq_add64(float acc, float term)
{
#pragma unroll
    for (int j = 0; j < 64; j++) acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(term), j));
    return acc;
}

}  // namespace mislam
