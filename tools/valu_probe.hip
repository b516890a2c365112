// Calibration probe for the `issue` rooflines (bench.py): kernels that do nothing but issue vector instructions of a KNOWN count, at
// 8 waves per SIMD on every CU -- run under `rocprofv3 --pmc` with the counter sets the search kernel is profiled with, their
// SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU and GRBM_GUI_ACTIVE say what those counters read at a (near) saturated vector pipe, and their
// event-timed rate is the practical peak the search kernel's instruction rate is divided by.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/valu_probe.hip -o tools/valu_probe;  tools/valu_probe
// (-fno-slp-vectorize: left to itself the compiler fuses the eight v_fma_f32 chains of probe 0 into four v_pk_fma_f32)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: v_fma_f32, 1: v_pk_fma_f32.  8 independent chains per lane.
template <int MODE>
__global__ __launch_bounds__(256) void valu_probe(float* out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f32x2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    const f32x2 pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; i++) {
        if constexpr (MODE == 0) {
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else {
            p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
            p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
            p4 = __builtin_elementwise_fma(p4, pa, pb); p5 = __builtin_elementwise_fma(p5, pa, pb);
            p6 = __builtin_elementwise_fma(p6, pa, pb); p7 = __builtin_elementwise_fma(p7, pa, pb);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                          p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}

template <int MODE>
static void run(const char* name, int cus, int instr_per_iter)
{
    const int iters = 20000, blocks = cus * 8;          // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    float* out;
    if (hipMalloc(&out, sizeof(float) * 256 * blocks) != hipSuccess) exit(2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(valu_probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(valu_probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * (double)iters * instr_per_iter;
    printf("{\"probe\": \"%s\", \"waves_per_simd\": 8, \"wave_instructions\": %.0f, \"ms\": %.4f, \"wave_instructions_per_s\": %.4g}\n", name, wave_instr, ms,
           wave_instr / (ms * 1e-3));
    hipFree(out);
}

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 2;
    run<0>("v_fma_f32", prop.multiProcessorCount, 8);
    run<1>("v_pk_fma_f32", prop.multiProcessorCount, 8);
    return 0;
}
