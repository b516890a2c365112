// Minimal reproducer for the exit-time SIGSEGV under rocprofv3: a process that makes ONE cooperative launch and returns
// from main -- nothing of libmislam in it.  Modes: plain (ordinary launch only) | coop (hipLaunchCooperativeKernel on the null stream) |
// coop_stream (on an own stream, destroyed before return) | coop_reset (hipDeviceReset() before return).  argv[2]: where to copy /proc/self/maps
// (so that the abort's frames can be given names: library + offset).
//   hipcc --offload-arch=gfx950 -O2 tools/coop_exit_probe.hip -o tools/coop_exit_probe
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstring>

__global__ void probe_kernel(int* out)
{
    if (threadIdx.x == 0) atomicAdd(out, 1);
}

__global__ void probe_coop_kernel(int* out)
{
    cooperative_groups::grid_group g = cooperative_groups::this_grid();
    if (threadIdx.x == 0) atomicAdd(out, 1);
    g.sync();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = out[0];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "coop";
    int* d = nullptr;
    CK(hipMalloc(&d, 2 * sizeof(int)));
    CK(hipMemset(d, 0, 2 * sizeof(int)));
    hipStream_t s = nullptr;
    if (!strcmp(mode, "coop_stream")) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (!strcmp(mode, "plain")) {
        probe_kernel<<<8, 64, 0, s>>>(d);
    } else {
        void* args[] = {&d};
        CK(hipLaunchCooperativeKernel(reinterpret_cast<void*>(probe_coop_kernel), dim3(8), dim3(64), args, 0, s));
    }
    CK(hipStreamSynchronize(s));
    int h[2] = {0, 0};
    CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    CK(hipFree(d));
    if (s) CK(hipStreamDestroy(s));
    if (!strcmp(mode, "coop_reset")) CK(hipDeviceReset());
    printf("%s: counted %d / %d\n", mode, h[0], h[1]);
    if (argc > 2) {
        FILE* in = fopen("/proc/self/maps", "r");
        FILE* out = fopen(argv[2], "w");
        char line[1024];
        while (in && out && fgets(line, sizeof line, in))
            if (strstr(line, " r-xp ") || strstr(line, ".so")) fputs(line, out);
        if (in) fclose(in);
        if (out) fclose(out);
    }
    fflush(stdout);
    return 0;
}
