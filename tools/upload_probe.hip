// Developer probe: host -> device upload of pageable memory through a pinned ring, staged by 1 or 2 host threads (tools/README.md).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int run(size_t PIECE)
{
    const size_t bytes = 12u << 20, SLOTS = 16;
    printf("piece %zu KB\n", PIECE >> 10);
    std::vector<char> src(bytes, 1);
    char *pin, *dev;
    hipHostMalloc((void**)&pin, PIECE * SLOTS, hipHostMallocDefault);
    hipMalloc((void**)&dev, bytes);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t ev[SLOTS];
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    // helper thread: copies the piece posted in `job` (sequence number in `posted`), acknowledges in `done`
    struct Job { const char* src; char* dst; size_t n; } job{};
    std::atomic<unsigned> posted{0}, done{0};
    std::atomic<bool> quit{false};
    std::thread helper([&] {
        unsigned seen = 0;
        while (!quit.load(std::memory_order_acquire)) {
            const unsigned p = posted.load(std::memory_order_acquire);
            if (p == seen) { __builtin_ia32_pause(); continue; }
            memcpy(job.dst, job.src, job.n);
            seen = p;
            done.store(p, std::memory_order_release);
        }
    });
    hipStream_t s2; hipStreamCreate(&s2);
    for (int mode = 0; mode < 4; mode++) {
        double best = 1e9;
        for (int rep = 0; rep < 8; rep++) {
            hipStreamSynchronize(s);
            const double t0 = now();
            if (mode == 0) {
                unsigned k = 0;
                for (size_t o = 0; o < bytes; o += PIECE, k++) {
                    char* slot = pin + (k % SLOTS) * PIECE;
                    if (k >= SLOTS) hipEventSynchronize(ev[k % SLOTS]);
                    memcpy(slot, src.data() + o, std::min(PIECE, bytes - o));
                    hipMemcpyAsync(dev + o, slot, std::min(PIECE, bytes - o), hipMemcpyHostToDevice, s);
                    hipEventRecord(ev[k % SLOTS], s);
                }
            } else if (mode == 1) {
                unsigned k = 0;
                for (size_t o = 0; o < bytes; o += 2 * PIECE, k += 2) {
                    char* a = pin + (k % SLOTS) * PIECE; char* b = pin + ((k + 1) % SLOTS) * PIECE;
                    if (k >= SLOTS) { hipEventSynchronize(ev[k % SLOTS]); hipEventSynchronize(ev[(k + 1) % SLOTS]); }
                    const bool two = o + PIECE < bytes;
                    unsigned ticket = 0;
                    if (two) { job = {src.data() + o + PIECE, b, PIECE}; ticket = posted.load() + 1; posted.store(ticket, std::memory_order_release); }
                    memcpy(a, src.data() + o, PIECE);
                    hipMemcpyAsync(dev + o, a, PIECE, hipMemcpyHostToDevice, s);
                    hipEventRecord(ev[k % SLOTS], s);
                    if (two) {
                        while (done.load(std::memory_order_acquire) != ticket) __builtin_ia32_pause();
                        hipMemcpyAsync(dev + o + PIECE, b, PIECE, hipMemcpyHostToDevice, s);
                        hipEventRecord(ev[(k + 1) % SLOTS], s);
                    }
                }
            } else if (mode == 2) {
                hipMemcpyAsync(dev, src.data(), bytes, hipMemcpyHostToDevice, s);
            } else {
                unsigned k = 0;
                for (size_t o = 0; o < bytes; o += PIECE, k++) {
                    char* slot = pin + (k % SLOTS) * PIECE;
                    if (k >= SLOTS) hipEventSynchronize(ev[k % SLOTS]);
                    memcpy(slot, src.data() + o, std::min(PIECE, bytes - o));
                    hipStream_t st = (k & 1) ? s2 : s;
                    hipMemcpyAsync(dev + o, slot, std::min(PIECE, bytes - o), hipMemcpyHostToDevice, st);
                    hipEventRecord(ev[k % SLOTS], st);
                }
                hipStreamSynchronize(s2);
            }
            const double t1 = now();
            hipStreamSynchronize(s);
            const double t2 = now();
            if (t2 - t0 < best) best = t2 - t0;
            if (rep == 7) printf("mode %d (%s): best %.3f ms (%.1f GB/s), last: host part %.3f ms, drain %.3f ms\n", mode,
                                 mode == 0 ? "ring, one thread" : mode == 1 ? "ring, two threads" : mode == 2 ? "runtime pageable copy" : "ring, two streams", best * 1e3, bytes / best / 1e9, (t1 - t0) * 1e3, (t2 - t1) * 1e3);
        }
    }
    quit.store(true);
    helper.join();
    return 0;
}
int main() { run(512u << 10); run(1u << 20); run(2u << 20); return 0; }
