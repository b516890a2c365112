// Stable LSD radix sort of (30-bit key, int value) pairs, 10 bits per pass -- the Hilbert orders of the two clouds (index build, at
// load time).  Own kernels: the library has no external device-library call left.
//
// One pass = a stable counting sort on a 10-bit digit, the scheme of the FGT member lists (cpd_fgt.hip): the array is cut into G
// chunks of consecutive elements, one wave each; (1) per-chunk digit counts H[g][d]; (2) per digit, the exclusive prefix of the
// counts over the chunks, and the digit starts; (3) every wave places its chunk, 64 elements per step in order: the lanes holding
// the same digit are found with ballots, ranked by lane, and take the next slots of that digit's run (cursors in LDS).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace mislam {

constexpr int RADIX_BITS = 10, RADIX_BUCKETS = 1 << RADIX_BITS;
constexpr int RADIX_MAX_CHUNKS = 4096;                     // G * 1024 counters <= 2^22 (what the scratch is sized for)
constexpr int RADIX_TARGET_CHUNKS = 1024;                  // chunks per pass: one wave each, ~1 000 elements at 1e6 -- the counter matrix stays at 4 MB

template <bool SCATTER>
__global__ __launch_bounds__(64) void radix_pass_kernel(const unsigned int* __restrict__ keys, const int* __restrict__ vals, int n, int shift,
                                                        int chunk, int* __restrict__ H, const int* __restrict__ off,
                                                        unsigned int* __restrict__ keys_out, int* __restrict__ vals_out)
{
    __shared__ int cur[RADIX_BUCKETS];
    const int lane = threadIdx.x;
    const int lo = blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    int* __restrict__ row = H + (size_t)blockIdx.x * RADIX_BUCKETS;
    for (int d = lane; d < RADIX_BUCKETS; d += 64) cur[d] = SCATTER ? off[d] + row[d] : 0;   // one wave: its LDS operations complete in order
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int i = i0 + lane;
        const bool valid = i < hi;
        const unsigned int key = valid ? keys[i] : 0u;
        const int val = valid ? vals[i] : 0;
        const int dig = (int)((key >> shift) & (RADIX_BUCKETS - 1));
        // the lanes holding the same digit: one ballot per digit BIT (ten) -- not one round per distinct digit, of which a step's 64
        // elements have ~60 (round 4; the member lists of cpd_fgt.hip learnt it first)
        unsigned long long same = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int b = 0; b < RADIX_BITS; b++) {
            const bool bit = ((dig >> b) & 1) != 0;
            const unsigned long long with = __builtin_amdgcn_ballot_w64(bit);
            same &= bit ? with : ~with;
        }
        const int cnt = (int)__builtin_popcountll(same);
        const int rank = (int)__builtin_popcountll(same & ((1ull << lane) - 1ull));
        int before = 0;
        if (valid && rank == 0) { before = cur[dig]; cur[dig] = before + cnt; }     // (the leaders of a step hold distinct digits)
        if (SCATTER) {
            before = __shfl(before, valid ? __builtin_ctzll(same) : lane, 64);
            if (valid) {
                const int pos = before + rank;
                keys_out[pos] = key;
                vals_out[pos] = val;
            }
        }
    }
    if (!SCATTER)
        for (int d = lane; d < RADIX_BUCKETS; d += 64) row[d] = cur[d];
}

// per digit: H[g][d] <- sum of the counts of the chunks before g; tot[d] <- the digit's total (one wave per digit)
__global__ __launch_bounds__(256) void radix_columns_kernel(int* __restrict__ H, int G, int* __restrict__ tot)
{
    const int d = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (d >= RADIX_BUCKETS) return;
    int carry = 0;
    for (int g0 = 0; g0 < G; g0 += 64) {
        const int g = g0 + lane;
        const int c = g < G ? H[(size_t)g * RADIX_BUCKETS + d] : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (g < G) H[(size_t)g * RADIX_BUCKETS + d] = carry + incl - c;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) tot[d] = carry;
}

// off[d] <- exclusive prefix of tot over the 1024 digits (one workgroup)
__global__ __launch_bounds__(RADIX_BUCKETS) void radix_offsets_kernel(const int* __restrict__ tot, int* __restrict__ off)
{
    __shared__ int s[RADIX_BUCKETS];
    const int v = tot[threadIdx.x];
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < RADIX_BUCKETS; o <<= 1) {
        const int t = (int)threadIdx.x >= o ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    off[threadIdx.x] = s[threadIdx.x] - v;
}

size_t radix_sort_temp_bytes(int) { return sizeof(int) * ((size_t)RADIX_MAX_CHUNKS * RADIX_BUCKETS + 2 * RADIX_BUCKETS); }

// keys_in/vals_in -> keys_out/vals_out, ascending by the low `bits` (<= 30, a multiple of 10) of the keys, stable.  Both pairs of
// arrays are used as ping-pong buffers; temp = radix_sort_temp_bytes().
hipError_t radix_sort_pairs_u32(void* temp, unsigned int* keys_in, unsigned int* keys_out, int* vals_in, int* vals_out, int n, int bits,
                                hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    int chunk = 64, G = (n + chunk - 1) / chunk;
    while (G > RADIX_TARGET_CHUNKS) { chunk += 64; G = (n + chunk - 1) / chunk; }   // (G x 1 024 counters cross memory three times per pass)
    int* H = reinterpret_cast<int*>(temp);
    int* tot = H + (size_t)G * RADIX_BUCKETS;
    int* off = tot + RADIX_BUCKETS;
    unsigned int *ki = keys_in, *ko = keys_out;
    int *vi = vals_in, *vo = vals_out;
    const int passes = (bits + RADIX_BITS - 1) / RADIX_BITS;
    for (int p = 0; p < passes; p++) {
        hipLaunchKernelGGL(radix_pass_kernel<false>, dim3(G), dim3(64), 0, s, ki, vi, n, p * RADIX_BITS, chunk, H, nullptr, nullptr, nullptr);
        hipLaunchKernelGGL(radix_columns_kernel, dim3(RADIX_BUCKETS / 4), dim3(256), 0, s, H, G, tot);
        hipLaunchKernelGGL(radix_offsets_kernel, dim3(1), dim3(RADIX_BUCKETS), 0, s, tot, off);
        hipLaunchKernelGGL(radix_pass_kernel<true>, dim3(G), dim3(64), 0, s, ki, vi, n, p * RADIX_BITS, chunk, H, off, ko, vo);
        unsigned int* tk = ki; ki = ko; ko = tk;
        int* tv = vi; vi = vo; vo = tv;
    }
    if (ki != keys_out) {                                      // an even number of passes left the result in the input arrays
        hipError_t e = hipMemcpyAsync(keys_out, ki, sizeof(unsigned int) * (size_t)n, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(vals_out, vi, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

}  // namespace mislam
