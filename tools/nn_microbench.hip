// Developer tool (not part of the product or the tests): times the K1 variants and two VALU-rate probes on the GPU box.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I cuda-slam_amd/csrc tools/nn_microbench.hip \
//         cuda-slam_amd/csrc/nn_kernel.hip -o tools/nn_microbench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "kernels.h"

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

using namespace mislam;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- VALU rate probes: ITER x 8 independent chains per lane
template <int MODE>
__global__ __launch_bounds__(256) void valu_probe(float* out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f32x2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    const f32x2 pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; i++) {
        if constexpr (MODE == 0) {  // scalar fma
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else if constexpr (MODE == 1) {  // packed fma
            p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
            p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
            p4 = __builtin_elementwise_fma(p4, pa, pb); p5 = __builtin_elementwise_fma(p5, pa, pb);
            p6 = __builtin_elementwise_fma(p6, pa, pb); p7 = __builtin_elementwise_fma(p7, pa, pb);
        } else if constexpr (MODE == 2) {  // packed add
            p0 = p0 + pa; p1 = p1 + pa; p2 = p2 + pa; p3 = p3 + pa; p4 = p4 + pa; p5 = p5 + pa; p6 = p6 + pa; p7 = p7 + pa;
        } else {  // scalar add
            x0 += a; x1 += a; x2 += a; x3 += a; x4 += a; x5 += a; x6 += a; x7 += a;
        }
    }
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y +
              p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
static void run_probe(const char* name, int blocks_per_cu)
{
    const int iters = 20000, blocks = 256 * blocks_per_cu;
    float* out;
    CK(hipMalloc(&out, sizeof(float) * 256 * blocks));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(valu_probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(valu_probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double lane_instr = (double)blocks * 256 * iters * 8;   // wave-instructions * 64
    printf("probe %-12s blocks/CU=%d  %.3f ms  %.3e lane-instr/s  (%.2f wave-instr/clk/CU @2.4GHz)\n", name, blocks_per_cu, ms,
           lane_instr / (ms * 1e-3), lane_instr / 64.0 / (ms * 1e-3) / 256.0 / 2.4e9);
    CK(hipFree(out));
}

static void cpu_nn(const std::vector<float>& s, int n, const std::vector<float>& t, int m, bool fma, std::vector<int>& idx,
                   std::vector<float>& d2)
{
    for (int i = 0; i < n; i++) {
        int bi = 0;
        float bd = INFINITY;
        for (int j = 0; j < m; j++) {
            const float dx = t[3 * j] - s[3 * i], dy = t[3 * j + 1] - s[3 * i + 1], dz = t[3 * j + 2] - s[3 * i + 2];
            const float d = fma ? fmaf(dz, dz, fmaf(dy, dy, dx * dx)) : (dx * dx + dy * dy) + dz * dz;
            if (d < bd) { bd = d; bi = j; }
        }
        idx[i] = bi;
        d2[i] = bd;
    }
}

struct Dev {
    float *sx, *sy, *sz, *tx, *ty, *tz;
    unsigned long long* keys;
    int n, n_pad, m, m_pad;
};

static Dev upload(const std::vector<float>& s, int n, const std::vector<float>& t, int m)
{
    Dev d;
    d.n = n; d.m = m;
    d.n_pad = (n + 2047) / 2048 * 2048;
    d.m_pad = m + 1024 * NN_TARGET_BLOCK * 64;   // generous: any chunking stays in bounds
    std::vector<float> h(std::max(d.n_pad, d.m_pad));
    float** sp[3] = {&d.sx, &d.sy, &d.sz};
    float** tp[3] = {&d.tx, &d.ty, &d.tz};
    for (int c = 0; c < 3; c++) {
        for (int i = 0; i < d.n_pad; i++) h[i] = s[3 * std::min(i, n - 1) + c];
        CK(hipMalloc(sp[c], sizeof(float) * d.n_pad));
        CK(hipMemcpy(*sp[c], h.data(), sizeof(float) * d.n_pad, hipMemcpyHostToDevice));
        for (int j = 0; j < d.m_pad; j++) h[j] = t[3 * std::min(j, m - 1) + c];
        CK(hipMalloc(tp[c], sizeof(float) * d.m_pad));
        CK(hipMemcpy(*tp[c], h.data(), sizeof(float) * d.m_pad, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&d.keys, sizeof(unsigned long long) * d.n_pad));
    return d;
}

static void free_dev(Dev& d)
{
    hipFree(d.sx); hipFree(d.sy); hipFree(d.sz); hipFree(d.tx); hipFree(d.ty); hipFree(d.tz); hipFree(d.keys);
}

static NnLaunch make_launch(const Dev& d, int R, int fma, int n_chunks)
{
    NnLaunch a{};
    a.sx = d.sx; a.sy = d.sy; a.sz = d.sz; a.n = d.n; a.n_pad = (d.n + 256 * R - 1) / (256 * R) * (256 * R);
    a.tx = d.tx; a.ty = d.ty; a.tz = d.tz;
    const int per = (d.m + n_chunks - 1) / n_chunks;
    a.chunk_len = (per + NN_TARGET_BLOCK - 1) / NN_TARGET_BLOCK * NN_TARGET_BLOCK;
    a.n_chunks = (d.m + a.chunk_len - 1) / a.chunk_len;
    a.index_base = 0; a.keys = d.keys; a.done_flag = nullptr; a.R = R; a.fma = fma;
    return a;
}

int main(int argc, char** argv)
{
    const int big = argc > 1 ? atoi(argv[1]) : 1000000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d  clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);

    for (int bpc : {1, 2, 4, 8}) {
        run_probe<0>("v_fma_f32", bpc);
        run_probe<1>("v_pk_fma_f32", bpc);
        run_probe<2>("v_pk_add_f32", bpc);
        run_probe<3>("v_add_f32", bpc);
    }

    std::mt19937 rng(666);
    std::uniform_real_distribution<float> U(-5.f, 5.f);

    // ---- correctness at 20k x 20k (with duplicates to exercise ties)
    {
        const int n = 20011, m = 19997;
        std::vector<float> s(3 * n), t(3 * m);
        for (auto& v : s) v = U(rng);
        for (auto& v : t) v = U(rng);
        for (int j = 0; j < m / 3; j++) memcpy(&t[3 * (m - 1 - j)], &t[3 * j], 12);          // duplicated targets
        for (int i = 0; i < 2000; i++) memcpy(&s[3 * i], &t[3 * ((i * 7) % m)], 12);          // exact hits
        Dev d = upload(s, n, t, m);
        for (int fma = 0; fma < 2; fma++) {
            std::vector<int> ridx(n);
            std::vector<float> rd2(n);
            cpu_nn(s, n, t, m, fma, ridx, rd2);
            for (int R : {1, 2, 4, 8})
                for (int chunks : {1, 7, 64}) {
                    NnLaunch a = make_launch(d, R, fma, chunks);
                    CK(hipMemset(d.keys, 0xFF, sizeof(unsigned long long) * d.n_pad));
                    CK(nn_launch(a, 0));
                    CK(hipDeviceSynchronize());
                    std::vector<unsigned long long> k(n);
                    CK(hipMemcpy(k.data(), d.keys, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
                    int bad = 0;
                    for (int i = 0; i < n; i++) {
                        const int gi = (int)(k[i] & 0xffffffffu);
                        const unsigned gb = (unsigned)(k[i] >> 32);
                        unsigned rb;
                        memcpy(&rb, &rd2[i], 4);
                        if (gi != ridx[i] || gb != rb) bad++;
                    }
                    printf("check fma=%d R=%d chunks=%d(%d x %d): %s (%d mismatches)\n", fma, R, chunks, a.n_chunks, a.chunk_len,
                           bad ? "FAIL" : "ok", bad);
                }
        }
        free_dev(d);
    }

    // ---- timing
    for (int n : {100000, big}) {
        const int m = n;
        std::vector<float> s(3 * (size_t)n), t(3 * (size_t)m);
        for (auto& v : s) v = U(rng);
        for (auto& v : t) v = U(rng);
        Dev d = upload(s, n, t, m);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int fma = 0; fma < 2; fma++)
            for (int R : {2, 4, 8})
                for (int chunks : {8, 16, 32, 64, 128}) {
                    NnLaunch a = make_launch(d, R, fma, chunks);
                    CK(hipMemset(d.keys, 0xFF, sizeof(unsigned long long) * d.n_pad));
                    CK(nn_launch(a, 0));
                    CK(hipDeviceSynchronize());
                    const int reps = n > 200000 ? 2 : 10;
                    CK(hipEventRecord(e0));
                    for (int r = 0; r < reps; r++) CK(nn_launch(a, 0));
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    ms /= reps;
                    const int wgs = a.n_pad / (256 * R) * a.n_chunks;
                    printf("N=M=%d fma=%d R=%d chunks=%d wgs=%d: %.3f ms  %.3e pairs/s\n", n, fma, R, a.n_chunks, wgs, ms,
                           (double)n * m / (ms * 1e-3));
                    fflush(stdout);
                }
        free_dev(d);
    }
    return 0;
}
