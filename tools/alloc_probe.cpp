// developer probe: host-to-device upload of fresh pageable buffers -- the runtime's own pageable path against a pinned staging
// buffer of our own and against pinning the caller's buffer in place
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t maxb = 16u << 20;
    void* d; hipMalloc(&d, maxb);
    char* pinned; hipHostMalloc((void**)&pinned, maxb, hipHostMallocDefault);
    std::vector<double> a, b, c;
    for (int rep = 0; rep < 40; rep++) {
        const size_t bytes = (size_t)(300000 + 25000 * rep) * 12;
        char* h = (char*)malloc(bytes); memset(h, rep, bytes);
        double t0 = now(); hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t1 = now();
        // own pinned staging, 1 MB pieces: the copy of piece k overlaps the DMA of piece k - 1
        const size_t piece = 1u << 20;
        for (size_t o = 0; o < bytes; o += piece) {
            const size_t nb = std::min(piece, bytes - o);
            memcpy(pinned + o, h + o, nb);
            hipMemcpyAsync((char*)d + o, pinned + o, nb, hipMemcpyHostToDevice, s);
        }
        hipStreamSynchronize(s); double t2 = now();
        hipHostRegister(h, bytes, hipHostRegisterDefault); double t3 = now();
        hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t4 = now();
        hipHostUnregister(h); double t5 = now();
        a.push_back(t1 - t0); b.push_back(t2 - t1); c.push_back(t5 - t2);
        printf("%zu B: runtime pageable %.3f ms | own pinned staging %.3f | register %.3f + copy %.3f + unregister %.3f\n", bytes, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4);
        free(h);
    }
    auto stat = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); printf("median %.3f max %.3f\n", v[v.size() / 2], v.back()); };
    stat(a); stat(b); stat(c);
    return 0;
}
