// K1 -- brute-force nearest-neighbour correspondence search for gfx950 (MI355X, wave64).
//
// Replaces FindCorrespondences (source/cuda-slam/cudacommon.cu:57-77; one thread per source point, serial loop over all
// targets through uncoalesced AoS global loads) and equals the search part of Common::GetCorrespondingPointsParallel
// (source/common/common.cpp:441-478): idx[i] = argmin_j |after[j] - before[i]|^2, strict '<', ascending j, so the lowest
// index wins ties.
//
// Design (all of it follows from "every lane of a wave wants the SAME target at the same time"):
//   * sources are register-blocked: each lane owns R source points (SoA loads, coalesced per component);
//   * targets are read with SCALAR loads (s_load_dwordx8 of the SoA x[], y[], z[] streams) and enter the VALU as SGPR
//     operands -- the wave-uniform broadcast is free, there is no LDS round trip and no VGPR spent on target data;
//   * two consecutive targets are evaluated per instruction with packed fp32 math (v_pk_add/mul/fma_f32): the SGPR pair
//     {t[j], t[j+1]} is one operand, the lane's source coordinate (broadcast to both halves) the other;
//   * the running minimum is kept WITHOUT per-pair index bookkeeping: a block of T targets only updates
//     m = min3(m, d_j, d_j+1); after the block a single compare per source tells whether the block improved the minimum,
//     and only then (rare: the expected number of improvements of a running minimum over K blocks is ~ln K) the wave
//     re-scans that block with the reference's sequential compare-and-select to find the index.  The result is exactly the
//     reference's (strict '<' between blocks keeps the earlier block on ties; the re-scan is the reference loop itself);
//   * the pair space is cut 2-D: blockIdx -> (source block, target chunk); partial results are merged with ONE 64-bit
//     atomicMin per source and chunk on the packed key (float_bits(d2) << 32 | global_target_index) -- d2 >= 0, so the
//     IEEE bit pattern orders like the value and the low word gives the lowest-index tie-break.  The same key is what the
//     multi-GPU path all-reduces with ncclMin.
//
// Arithmetic: IEEE fp32, no contraction unless FMA is asked for (compile this file with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace mislam {

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool FMA>
__device__ __forceinline__ f32x2 dist2_pk(f32x2 tx, f32x2 ty, f32x2 tz, f32x2 sx, f32x2 sy, f32x2 sz)
{
    const f32x2 dx = tx - sx, dy = ty - sy, dz = tz - sz;
    if constexpr (FMA) {
        return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
    } else {
        return (dx * dx + dy * dy) + dz * dz;
    }
}

template <bool FMA>
__device__ __forceinline__ float dist2_1(float tx, float ty, float tz, float sx, float sy, float sz)
{
    const float dx = tx - sx, dy = ty - sy, dz = tz - sz;
    if constexpr (FMA) {
        return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    } else {
        return (dx * dx + dy * dy) + dz * dz;
    }
}

// R sources per lane, T targets per block (T even).  Launch: 256 threads, grid = n_src_blocks * n_chunks.
// Preconditions (checked by the host launcher):
//   sx/sy/sz hold n_src_blocks*256*R floats (tail padded with copies of a real point),
//   tx/ty/tz hold >= n_chunks*chunk_len floats (tail padded with copies of the LAST real target: a duplicate can never
//   beat the original under strict '<'), chunk_len % T == 0.
template <int R, int T, bool FMA>
__global__ __launch_bounds__(256) void nn_bruteforce_kernel(
    const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, int n,
    const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz,
    int chunk_len, int n_chunks, int index_base, unsigned long long* __restrict__ keys, const int* __restrict__ done_flag)
{
    if (done_flag != nullptr && *done_flag != 0) return;

    // XCD-aware decomposition.  Consecutive block ids are dealt round-robin over the 8 XCDs, each with a private 4 MiB L2.
    // With n_chunks a multiple of 8, XCD x = id % 8 owns the chunks {x, x+8, ...} and walks them ONE AT A TIME: all source
    // blocks pass over chunk x before any block touches chunk x+8, so the target bytes an XCD is streaming (<= ~2 MB, see
    // plan_nn) stay resident in its L2 and are fetched from HBM/MALL once instead of once per source block.
    // Placement only affects speed/traffic, never the result; other chunk counts use the plain mapping.
    int chunk, sblk;
    if ((n_chunks & 7) == 0) {
        const int n_sblk = gridDim.x / n_chunks;
        const int local = blockIdx.x >> 3;
        chunk = ((local / n_sblk) << 3) + (blockIdx.x & 7);
        sblk = local % n_sblk;
    } else {
        chunk = blockIdx.x % n_chunks;
        sblk = blockIdx.x / n_chunks;
    }
    const int src0 = sblk * (256 * R) + threadIdx.x;

    f32x2 px[R], py[R], pz[R];
    float best[R];
    int bidx[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const float x = sx[src0 + r * 256], y = sy[src0 + r * 256], z = sz[src0 + r * 256];
        px[r] = f32x2{x, x};
        py[r] = f32x2{y, y};
        pz[r] = f32x2{z, z};
        // Starting bound: whatever key is already posted for this source is a REAL candidate (an earlier chunk's result,
        // or the previous ICP iteration's match re-evaluated under the new transform), so nothing farther away can be the
        // answer.  The bound is that distance plus one ulp: a target at exactly that distance but with a lower index must
        // still be found (the packed atomicMin then settles the tie).  A stale read only gives a looser, still valid bound.
        const unsigned int hi = (unsigned int)(__hip_atomic_load(&keys[src0 + r * 256 < n ? src0 + r * 256 : n - 1], __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT) >> 32);
        best[r] = hi < 0x7f800000u ? __uint_as_float(hi + 1u) : __builtin_inff();
        bidx[r] = -1;
    }

    const int j_begin = chunk * chunk_len;
    const float* __restrict__ cx = tx + j_begin;
    const float* __restrict__ cy = ty + j_begin;
    const float* __restrict__ cz = tz + j_begin;

    for (int j0 = 0; j0 < chunk_len; j0 += T) {
        float m[R];
#pragma unroll
        for (int r = 0; r < R; r++) m[r] = best[r];

#pragma unroll
        for (int k = 0; k < T; k += 2) {
            // wave-uniform addresses -> scalar loads; the pair lands in an aligned SGPR pair
            const f32x2 ax = f32x2{cx[j0 + k], cx[j0 + k + 1]};
            const f32x2 ay = f32x2{cy[j0 + k], cy[j0 + k + 1]};
            const f32x2 az = f32x2{cz[j0 + k], cz[j0 + k + 1]};
#pragma unroll
            for (int r = 0; r < R; r++) {
                const f32x2 d = dist2_pk<FMA>(ax, ay, az, px[r], py[r], pz[r]);
                m[r] = __builtin_fminf(__builtin_fminf(m[r], d.x), d.y);
            }
        }

        bool improved = false;
#pragma unroll
        for (int r = 0; r < R; r++) improved |= (m[r] < best[r]);

        if (__builtin_amdgcn_ballot_w64(improved) != 0ull) {
            // rare path: the reference's own sequential scan over this block
            for (int k = 0; k < T; k++) {
                const float ax = cx[j0 + k], ay = cy[j0 + k], az = cz[j0 + k];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const float d = dist2_1<FMA>(ax, ay, az, px[r].x, py[r].x, pz[r].x);
                    if (d < best[r]) { best[r] = d; bidx[r] = j_begin + j0 + k; }
                }
            }
        }
    }

#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = src0 + r * 256;
        if (i < n && bidx[r] >= 0) {   // bidx < 0: nothing in this chunk reached the starting bound
            const unsigned long long key =
                ((unsigned long long)__float_as_uint(best[r]) << 32) | (unsigned int)(bidx[r] + index_base);
            atomicMin(&keys[i], key);
        }
    }
}

#define MI_NN_INSTANTIATE(R, T)                                                                                        \
    template __global__ void nn_bruteforce_kernel<R, T, false>(const float*, const float*, const float*, int,          \
        const float*, const float*, const float*, int, int, int, unsigned long long*, const int*);                     \
    template __global__ void nn_bruteforce_kernel<R, T, true>(const float*, const float*, const float*, int,           \
        const float*, const float*, const float*, int, int, int, unsigned long long*, const int*);

MI_NN_INSTANTIATE(8, 16)
MI_NN_INSTANTIATE(4, 16)
MI_NN_INSTANTIATE(2, 16)
MI_NN_INSTANTIATE(1, 16)

template <int R, bool FMA>
static hipError_t launch_R(const NnLaunch& a, hipStream_t stream)
{
    constexpr int T = NN_TARGET_BLOCK;
    const int n_src_blocks = a.n_pad / (256 * R);
    dim3 grid((unsigned)(n_src_blocks * a.n_chunks)), block(256);
    hipLaunchKernelGGL((nn_bruteforce_kernel<R, T, FMA>), grid, block, 0, stream, a.sx, a.sy, a.sz, a.n, a.tx, a.ty, a.tz,
                       a.chunk_len, a.n_chunks, a.index_base, a.keys, a.done_flag);
    return hipGetLastError();
}

hipError_t nn_launch(const NnLaunch& a, hipStream_t stream)
{
    if (a.n <= 0 || a.n_chunks <= 0) return hipSuccess;
    switch (a.R) {
    case 8: return a.fma ? launch_R<8, true>(a, stream) : launch_R<8, false>(a, stream);
    case 4: return a.fma ? launch_R<4, true>(a, stream) : launch_R<4, false>(a, stream);
    case 2: return a.fma ? launch_R<2, true>(a, stream) : launch_R<2, false>(a, stream);
    case 1: return a.fma ? launch_R<1, true>(a, stream) : launch_R<1, false>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mislam

// Touching one kernel of this translation unit makes the runtime load its code object now (mi_ctx_create) instead of at the
// first launch inside a registration call (deferred loading: 5-16 ms per object, once).
namespace mislam {
__global__ void preload_nn_kernel_kernel() {}
hipError_t preload_nn_kernel()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(preload_nn_kernel_kernel));
}
}  // namespace mislam
