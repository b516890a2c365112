// developer probe: what one candidate test of the grid scan costs the vector pipe, by formulation (8 waves per SIMD on every CU, candidates in
// registers: no memory).  Prints ns per candidate per wave-slot and the implied cycles per candidate per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/cand_probe.hip -o tools/cand_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// MODE 0: distance only (min).  1: u64 key compare + 3 selects (the scan's form).  2: f32 compares (lt | eq & idx lt) + 3 selects.
// 3: u64 key via 32-bit pieces: hi compare + lo compare.  4: distance + f32 lt only + 2 selects (no tie rule: not exact).
template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters, float q0, float q1, float q2)
{
    float x = threadIdx.x * 0.37f, y = blockIdx.x * 0.11f, z = 1.5f;
    unsigned int w = threadIdx.x;
    unsigned long long kbest = ~0ull;
    float best = 3.4e38f;
    unsigned int bidx = ~0u, bslot = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            x += 0.001f; y -= 0.002f; w += 7u;                       // (stand-ins for freshly loaded candidates: 3 cheap ops)
            const float dx = x - q0, dy = y - q1, dz = z - q2;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if constexpr (MODE == 0) { best = fminf(best, d); }
            else if constexpr (MODE == 1) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | w;
                const bool better = key < kbest;
                kbest = better ? key : kbest; bslot = better ? (unsigned int)(i * 4 + j) : bslot;
            } else if constexpr (MODE == 2) {
                const bool better = (d < best) | ((d == best) & (w < bidx));
                best = better ? d : best; bidx = better ? w : bidx; bslot = better ? (unsigned int)(i * 4 + j) : bslot;
            } else if constexpr (MODE == 3) {
                const unsigned int hi = __float_as_uint(d), bh = (unsigned int)(kbest >> 32), bl = (unsigned int)kbest;
                const bool better = (hi < bh) | ((hi == bh) & (w < bl));
                kbest = better ? (((unsigned long long)hi << 32) | w) : kbest; bslot = better ? (unsigned int)(i * 4 + j) : bslot;
            } else {
                const bool better = d < best;
                best = better ? d : best; bslot = better ? (unsigned int)(i * 4 + j) : bslot;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = kbest + (unsigned long long)__float_as_uint(best) + bidx + bslot;
}

// MODE 5: candidates as PAIRS (x0,x1) (y0,y1) (z0,z1) (w0,w1) -- the distance of two candidates per packed instruction -- + u64 key
// compare + 2 selects (no slot).  MODE 6: the same unpacked (one candidate per instruction), 2 selects.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void probe_pairs(unsigned long long* out, int iters, float q0, float q1, float q2)
{
    f2 X = {threadIdx.x * 0.37f, threadIdx.x * 0.39f}, Y = {blockIdx.x * 0.11f, blockIdx.x * 0.13f}, Z = {1.5f, 1.6f};
    u2 W = {threadIdx.x, threadIdx.x + 1u};
    const f2 Q0 = {q0, q0}, Q1 = {q1, q1}, Q2 = {q2, q2};
    unsigned long long kbest = ~0ull;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            X += (f2){0.001f, 0.002f}; Y -= (f2){0.002f, 0.001f}; W += (u2){7u, 9u};      // (stand-ins for a freshly loaded pair)
            if constexpr (MODE == 5) {
                const f2 dx = X - Q0, dy = Y - Q1, dz = Z - Q2;
                const f2 d = (dx * dx + dy * dy) + dz * dz;
                const unsigned long long k0 = ((unsigned long long)__float_as_uint(d.x) << 32) | W.x, k1 = ((unsigned long long)__float_as_uint(d.y) << 32) | W.y;
                kbest = k0 < kbest ? k0 : kbest;
                kbest = k1 < kbest ? k1 : kbest;
            } else {
                const float dx0 = X.x - q0, dy0 = Y.x - q1, dz0 = Z.x - q2, dx1 = X.y - q0, dy1 = Y.y - q1, dz1 = Z.y - q2;
                float d0 = (dx0 * dx0 + dy0 * dy0) + dz0 * dz0, d1 = (dx1 * dx1 + dy1 * dy1) + dz1 * dz1;
                asm volatile("" : "+v"(d0), "+v"(d1));                // (keeps the vectoriser from pairing them up again)
                const unsigned long long k0 = ((unsigned long long)__float_as_uint(d0) << 32) | W.x, k1 = ((unsigned long long)__float_as_uint(d1) << 32) | W.y;
                kbest = k0 < kbest ? k0 : kbest;
                kbest = k1 < kbest ? k1 : kbest;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = kbest;
}

template <int MODE>
static void run(const char* name, int cus)
{
    const int iters = 20000, blocks = cus * 8;
    unsigned long long* out;
    if (hipMalloc(&out, sizeof(unsigned long long) * 256 * blocks) != hipSuccess) exit(2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    if constexpr (MODE >= 5) hipLaunchKernelGGL(probe_pairs<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 0.1f, 0.2f, 0.3f);
    else hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 0.1f, 0.2f, 0.3f);
    hipEventRecord(e0, 0);
    if constexpr (MODE >= 5) hipLaunchKernelGGL(probe_pairs<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.1f, 0.2f, 0.3f);
    else hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.1f, 0.2f, 0.3f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double cand_per_simd = 8.0 * iters * 4;                    // 8 waves per SIMD
    printf("%-44s %.3f ms  -> %.1f cycles per candidate per SIMD (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / cand_per_simd);
    hipFree(out);
}

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 2;
    run<0>("distance + min", prop.multiProcessorCount);
    run<1>("distance + u64 key compare + 3 selects", prop.multiProcessorCount);
    run<2>("distance + f32 lt | eq & idx lt + 3 selects", prop.multiProcessorCount);
    run<3>("distance + 32-bit pieces of the key", prop.multiProcessorCount);
    run<4>("distance + f32 lt + 2 selects (no tie rule)", prop.multiProcessorCount);
    run<5>("PAIRS: packed distance + u64 key + 2 selects", prop.multiProcessorCount);
    run<6>("pairs, unpacked distance + u64 key + 2 selects", prop.multiProcessorCount);
    return 0;
}
