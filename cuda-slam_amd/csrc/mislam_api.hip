// C ABI (include/mi_slam.h): context, ICP driver and the test-grade ICP primitives.
//
// The ICP driver is the MI355X-native counterpart of CudaICP (source/cuda-slam/icpcuda.cu:8-58): the same loop, but the
// whole iteration -- search, (all-reduce), moments, solve, transform, error, stop rule -- is enqueued on one HIP stream
// without a single host round trip; the host only reads the 256-byte state block back every `sync_every` iterations.
#include <hip/hip_runtime.h>
#include <sys/resource.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "context.h"

using namespace mislam;

// ---------------------------------------------------------------------------------------------------------------
// error reporting
// ---------------------------------------------------------------------------------------------------------------
namespace mislam {
static thread_local char g_error[512] = "";
void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}
}  // namespace mislam

namespace mislam {
static thread_local std::vector<void*>* t_retire_sink = nullptr;      // the running call's context list (CtxScope)
double& alloc_ms_counter()
{
    static thread_local double ms = 0.0;
    return ms;
}
double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void device_free(void* p);
void retire_later(void* p)
{
    if (t_retire_sink != nullptr) { t_retire_sink->push_back(p); return; }
    (void)hipDeviceSynchronize();               // outside any context call (not a path the library takes): nothing may still read it after this
    device_free(p);
}
CtxScope::CtxScope(mi_ctx* c) : ctx(c), outer(t_retire_sink) { t_retire_sink = &c->retired; }
CtxScope::~CtxScope()
{
    t_retire_sink = outer;
    if (!ctx->retired.empty() && ctx->stream != nullptr && hipStreamQuery(ctx->stream) == hipSuccess) retire_buffers(ctx);
    else (void)hipGetLastError();               // (hipErrorNotReady is not an error here)
}
// One allocation stream and one PRIVATE memory pool per device (hipMemPoolCreate; release threshold: never -- the library's buffers
// are meant to be handed out again, not returned to the driver between calls).  The device's default pool is left alone: its
// attributes belong to the host application.  The stream carries no work, so the pool's stream-ordered calls complete at once
// there.  A buffer is only ever freed behind a synchronisation of the stream that used it (retire_buffers, context destruction),
// so handing its memory out again -- to this context or another -- is safe whatever stream the new owner works on.  When the last
// context on a device is destroyed the pool is trimmed to nothing (pool_context_gone).  MISLAM_POOL=0: plain hipMalloc / hipFree.
static constexpr int MAX_DEVICES = 64;
static hipStream_t g_alloc_stream[MAX_DEVICES] = {nullptr};
static hipMemPool_t g_pool[MAX_DEVICES] = {nullptr};
static int g_pool_contexts[MAX_DEVICES] = {0};
static int g_use_pool = -1;                     // -1: not decided yet
static std::mutex g_pool_mutex;
static hipStream_t alloc_stream(hipMemPool_t* pool_out = nullptr)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (g_use_pool == -1) {
        const char* e = getenv("MISLAM_POOL");
        g_use_pool = (e && e[0] == '0') ? 0 : 1;
    }
    if (!g_use_pool) return nullptr;
    if (g_alloc_stream[dev] == nullptr) {
        hipMemPool_t pool = nullptr;
        hipStream_t s = nullptr;
        unsigned long long keep = ~0ull;
        hipMemPoolProps props;
        memset(&props, 0, sizeof props);
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = dev;
        if (hipMemPoolCreate(&pool, &props) != hipSuccess || hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) != hipSuccess ||
            hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            if (pool != nullptr) (void)hipMemPoolDestroy(pool);
            g_use_pool = 0;
            return nullptr;
        }
        g_pool[dev] = pool;
        g_alloc_stream[dev] = s;
    }
    if (pool_out) *pool_out = g_pool[dev];
    return g_alloc_stream[dev];
}
static bool g_pool_proven = false;               // a pool allocation has succeeded: the mode never changes after that
hipError_t device_alloc(void** p, size_t bytes)
{
    hipMemPool_t pool = nullptr;
    if (hipStream_t s = alloc_stream(&pool)) {
        hipError_t e = hipMallocFromPoolAsync(p, bytes, pool, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);          // (nothing else is ever on this stream: the memory is usable on any stream from here)
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        if (e == hipSuccess) { g_pool_proven = true; return e; }
        (void)hipGetLastError();
        if (g_pool_proven || e == hipErrorOutOfMemory) return e;
        g_use_pool = 0;                          // this runtime has no working pool: plain allocations throughout
    }
    return hipMalloc(p, bytes);
}
void device_free(void* p)
{
    if (p == nullptr) return;
    if (hipStream_t s = alloc_stream()) { (void)hipFreeAsync(p, s); return; }
    (void)hipFree(p);
}
// context bookkeeping per device: the last one to go returns the pool's memory to the driver
static void pool_context_created(int dev)
{
    if (dev < 0 || dev >= MAX_DEVICES) return;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_pool_contexts[dev] += 1;
}
static void pool_context_gone(int dev)
{
    if (dev < 0 || dev >= MAX_DEVICES) return;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (g_pool_contexts[dev] > 0) g_pool_contexts[dev] -= 1;
    if (g_pool_contexts[dev] == 0 && g_pool[dev] != nullptr && g_alloc_stream[dev] != nullptr) {
        (void)hipStreamSynchronize(g_alloc_stream[dev]);           // the frees enqueued by the destructor have landed
        (void)hipMemPoolTrimTo(g_pool[dev], 0);
    }
}
// (called right behind a synchronisation of the context's stream, its device current: nothing still uses these)
void retire_buffers(mi_ctx* ctx)
{
    std::vector<void*> v;
    v.swap(ctx->retired);
    for (void* p : v) device_free(p);
}
}  // namespace mislam

// developer switch MISLAM_DEV_STALL_MS=<ms>: report any host-side section that takes longer, with the calling thread's context switches
// over it -- an INVOLUNTARY one with no voluntary ones means the thread was taken off its core (a CPU quota of the container
// running out: numpy's BLAS pool spinning on every host core did exactly that to the round-2 sweeps), not that the GPU or the
// runtime made it wait.
static double g_stall_ms = -1.0;
struct StallProbe {
    const char* label; double t0; struct rusage ru0;
    explicit StallProbe(const char* l) : label(l), t0(0.0) { if (g_stall_ms > 0) { t0 = mislam::wall_ms(); getrusage(RUSAGE_THREAD, &ru0); } }
    ~StallProbe()
    {
        if (g_stall_ms <= 0) return;
        const double d = mislam::wall_ms() - t0;
        if (d <= g_stall_ms) return;
        struct rusage ru1; getrusage(RUSAGE_THREAD, &ru1);
        fprintf(stderr, "mislam stall: %s %.2f ms (context switches: %ld voluntary, %ld involuntary)\n", label, d, ru1.ru_nvcsw - ru0.ru_nvcsw, ru1.ru_nivcsw - ru0.ru_nivcsw);
    }
};

extern "C" const char* mi_last_error(void) { return g_error; }
extern "C" int mi_abi_version(void) { return MI_SLAM_ABI_VERSION; }

extern "C" int mi_device_count(int* count)
{
    if (!count) { set_error("mi_device_count: null argument"); return MI_ERR_INVALID_ARG; }
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0) {
        *count = 0;
        set_error("no usable HIP device (%s); this library has no CPU fallback", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return MI_ERR_NO_DEVICE;
    }
    *count = c;
    return MI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------------------
// Loads every code object of the library now.  With the runtime's deferred loading the first launch out of each translation
// unit otherwise stalls whatever call it happens in (measured: first ICP call at 1e6 points 22 -> 17 ms, first CPD call
// 36 -> 3 ms) -- at the price of ~170 ms here, so it is the caller's choice: a long-lived process wants it, a one-shot run not.
extern "C" int mi_ctx_preload(mi_ctx* c)
{
    if (!c) { set_error("mi_ctx_preload: null context"); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    MI_HIP(preload_nn_kernel()); MI_HIP(preload_nn_tree()); MI_HIP(preload_nn_grid()); MI_HIP(preload_icp_kernels()); MI_HIP(preload_cpd_kernels());
    MI_HIP(preload_cpd_fgt()); MI_HIP(preload_nicp_api()); MI_HIP(preload_prepare_api());
    return MI_OK;
}

static int ctx_create_common(int device, mi_ctx** out)
{
    if (!out) { set_error("mi_ctx_create: null out pointer"); return MI_ERR_INVALID_ARG; }
    *out = nullptr;
    int count = 0;
    MI_TRY(mi_device_count(&count));
    if (device < 0 || device >= count) { set_error("mi_ctx_create: device %d out of range [0,%d)", device, count); return MI_ERR_INVALID_ARG; }
    MI_HIP(hipSetDevice(device));
    mi_ctx* c = new mi_ctx();
    c->device = device;
    pool_context_created(device);
    // a failure half way releases what was created so far (mi_ctx_destroy copes with null members)
    const int rc = [&]() -> int {
        hipDeviceProp_t prop;
        MI_HIP(hipGetDeviceProperties(&prop, device));
        c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        MI_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        MI_HIP(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
        MI_HIP(hipStreamCreateWithFlags(&c->aux2, hipStreamNonBlocking));
        for (hipEvent_t& e : c->aux_event) MI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        MI_HIP(hipMalloc((void**)&c->d_state, sizeof(IcpState)));
        MI_HIP(hipHostMalloc((void**)&c->h_state, sizeof(IcpState), hipHostMallocDefault));
        MI_HIP(hipEventCreateWithFlags(&c->peek_event, hipEventDisableTiming));
        memset(c->h_state, 0, sizeof(IcpState));
        // developer switches: read here, once -- nothing on the per-iteration path looks at the environment
        auto env_i = [](const char* name, int dflt) { const char* v = getenv(name); return (v && *v) ? atoi(v) : dflt; };
        if (const char* sm = getenv("MISLAM_DEV_STALL_MS")) g_stall_ms = atof(sm);
        c->tune.nn_force_mode = env_i("MISLAM_NN_MODE", 0);
        c->tune.nn_R = env_i("MISLAM_NN_R", 2);
        c->tune.nn_wgs = env_i("MISLAM_NN_WGS", 0);
        c->tune.nn_chunks = env_i("MISLAM_NN_CHUNKS", 0);
        c->tune.cpd_mfma = env_i("MISLAM_CPD_MFMA", 1);
        c->tune.cpd_trunc_cull = env_i("MISLAM_CPD_TRUNC_CULL", 1);
        c->tune.fgt_resume = env_i("MISLAM_FGT_RESUME", 1);
        c->tune.fgt_replay = env_i("MISLAM_FGT_REPLAY", 1);
        c->tune.fgt_two_streams = env_i("MISLAM_FGT_TWO_STREAMS", 1);
        c->tune.fgt_lists_in_model = env_i("MISLAM_FGT_LISTS_IN_MODEL", 1);
        c->tune.fgt_coop_sweep = env_i("MISLAM_FGT_COOP_SWEEP", 1);
        c->tune.fgt_model_splits = env_i("MISLAM_FGT_MODEL_SPLITS", 1);
        c->tune.fgt_shard_queries = env_i("MISLAM_FGT_SHARD_QUERIES", 1);
        c->tune.grid_deal_rows = env_i("MISLAM_GRID_DEAL_ROWS", -1);
        c->tune.grid_split_walks = env_i("MISLAM_GRID_SPLIT_WALKS", -1);
        c->tune.icp_pipeline = env_i("MISLAM_ICP_PIPELINE", 1);
        c->tune.icp_fused_solve = env_i("MISLAM_ICP_FUSED_SOLVE", 1);
        c->tune.svd_ieee = env_i("MISLAM_SVD_IEEE", 0);
        if (const char* ppc = getenv("MISLAM_GRID_PPC")) { const float f = (float)atof(ppc); if (f >= 0.25f && f <= 64.f) c->tune.grid_points_per_cell = f; }
        if (env_i("MISLAM_PRELOAD", 0) == 1) MI_TRY(mi_ctx_preload(c));       // =1: mi_ctx_preload as part of every context creation
        // Host clouds go up through the runtime's own pageable-copy path (round 6).  Rounds 3-5 staged them through an own pinned ring of 16 x 1 MB, built
        // against 20-50 ms stalls that round 3 pinned on the runtime's path -- and that were the process's CPU quota being throttled by idle BLAS pools
        // (profiles/r03_stall_hunt.log: gone with the pools quiet, for either path).  Measured side by side, 20 calls each (profiles/r06_upload_paths.log):
        // a load of two 12 MB clouds 1.35 -> 1.01 ms (the ring's one host thread copies every byte itself, 37 us per MB; the runtime pipelines larger pieces
        // over its own staging buffers), 120 MB clouds 13.3 -> 7.8 ms, the whole 50-iteration registration at 1e6 points 6.43 -> 6.04 ms; outliers as
        // rare on one as on the other.  MISLAM_PIN=1 brings the ring back (4 ms of pinning per context).
        if (env_i("MISLAM_PIN", 0) != 0) {
            MI_HIP(hipHostMalloc((void**)&c->pin, mi_ctx::PIN_PIECE * mi_ctx::PIN_SLOTS, hipHostMallocDefault));
            for (hipEvent_t& e : c->pin_event) MI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        MI_HIP(hipHostMalloc((void**)&c->h_scratch, 64 * sizeof(float), hipHostMallocDefault));
        return MI_OK;
    }();
    if (rc != MI_OK) { mi_ctx_destroy(c); return rc; }
    *out = c;
    return MI_OK;
}

extern "C" int mi_ctx_create(int device, mi_ctx** out) { return ctx_create_common(device, out); }

extern "C" int mi_dist_unique_id(void* out_unique_id)
{
    if (!out_unique_id) { set_error("mi_dist_unique_id: null argument"); return MI_ERR_INVALID_ARG; }
    static_assert(sizeof(ncclUniqueId) == MI_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    MI_NCCL(ncclGetUniqueId(&id));
    memcpy(out_unique_id, &id, sizeof id);
    return MI_OK;
}

extern "C" int mi_ctx_create_dist(int device, int rank, int world, const void* unique_id, mi_ctx** out)
{
    if (world < 1 || rank < 0 || rank >= world || !unique_id) { set_error("mi_ctx_create_dist: bad rank/world/id"); return MI_ERR_INVALID_ARG; }
    MI_TRY(ctx_create_common(device, out));
    mi_ctx* c = *out;
    c->rank = rank;
    c->world = world;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, ncclGetErrorString(r));
        mi_ctx_destroy(c);
        *out = nullptr;
        return MI_ERR_RCCL;
    }
    return MI_OK;
}

extern "C" int mi_ctx_create_exchange(int device, int rank, int world, mi_exchange_fn exchange, void* user, mi_ctx** out)
{
    if (world < 1 || rank < 0 || rank >= world || !exchange) { set_error("mi_ctx_create_exchange: bad rank/world/callback"); return MI_ERR_INVALID_ARG; }
    MI_TRY(ctx_create_common(device, out));
    mi_ctx* c = *out;
    c->rank = rank;
    c->world = world;
    c->exchange = exchange;
    c->exchange_user = user;
    return MI_OK;
}

// The caller's transport: drain the stream, stage through pinned host memory, combine there, copy back.
static int exchange_on_host(mi_ctx* c, void* dev_ptr, size_t count, int kind)
{
    const size_t bytes = count * 8;
    if (bytes > c->exchange_cap) {
        if (c->exchange_host) (void)hipHostFree(c->exchange_host);
        c->exchange_host = nullptr;
        c->exchange_cap = 0;
        MI_HIP(hipHostMalloc(&c->exchange_host, bytes, hipHostMallocDefault));
        c->exchange_cap = bytes;
    }
    MI_HIP(hipMemcpyAsync(c->exchange_host, dev_ptr, bytes, hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipStreamSynchronize(c->stream));
    const int r = c->exchange(c->exchange_user, c->exchange_host, count, kind);
    if (r != 0) { set_error("the exchange callback failed with %d (rank %d of %d, %zu elements, kind %d)", r, c->rank, c->world, count, kind); return MI_ERR_RCCL; }
    MI_HIP(hipMemcpyAsync(dev_ptr, c->exchange_host, bytes, hipMemcpyHostToDevice, c->stream));
    return MI_OK;
}

int mislam::allreduce_min_u64(mi_ctx* c, unsigned long long* dev_ptr, size_t count)
{
    if (c->exchange) return exchange_on_host(c, dev_ptr, count, MI_EXCHANGE_MIN_U64);
    if (!c->comm) return MI_OK;
    MI_NCCL(ncclAllReduce(dev_ptr, dev_ptr, count, ncclUint64, ncclMin, c->comm, c->stream));
    return MI_OK;
}

int mislam::allreduce_sum_f64(mi_ctx* c, double* dev_ptr, size_t count)
{
    if (c->exchange) return exchange_on_host(c, dev_ptr, count, MI_EXCHANGE_SUM_F64);
    if (!c->comm) return MI_OK;
    MI_NCCL(ncclAllReduce(dev_ptr, dev_ptr, count, ncclDouble, ncclSum, c->comm, c->stream));
    return MI_OK;
}

extern "C" int mi_ctx_rank(const mi_ctx* ctx, int* rank, int* world)
{
    if (!ctx) { set_error("mi_ctx_rank: null context"); return MI_ERR_INVALID_ARG; }
    if (rank) *rank = ctx->rank;
    if (world) *world = ctx->world;
    return MI_OK;
}

extern "C" int mi_dist_info(mi_ctx* c, int* nranks, int* rank, unsigned long long* ranks_seen)
{
    if (!c) { set_error("mi_dist_info: null context"); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    int n = c->world, r = c->rank;
    if (c->comm) {
        MI_NCCL(ncclCommCount(c->comm, &n));
        MI_NCCL(ncclCommUserRank(c->comm, &r));
    }
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    if (ranks_seen) {
        if (c->world > 52) { set_error("mi_dist_info: ranks_seen holds at most 52 ranks"); return MI_ERR_INVALID_ARG; }
        // 2^rank is exact in a double, and so is any sum of distinct powers of two below 2^53: the SUM all-reduce of the moments
        // path carries the mask
        MI_TRY(c->rows_reduced.reserve(64 * 18));
        const double mine = (double)(1ull << c->rank);
        MI_HIP(hipMemcpyAsync(c->rows_reduced.p, &mine, sizeof mine, hipMemcpyHostToDevice, c->stream));
        MI_TRY(allreduce_sum_f64(c, c->rows_reduced.p, 1));
        double all = 0.0;
        MI_HIP(hipMemcpyAsync(&all, c->rows_reduced.p, sizeof all, hipMemcpyDeviceToHost, c->stream));
        MI_HIP(hipStreamSynchronize(c->stream));
        *ranks_seen = (unsigned long long)all;
    }
    return MI_OK;
}

extern "C" int mi_runtime_info(char* hip_path, char* rccl_path, int cap, int* hip_runtime_version, int* rccl_version)
{
    const auto object_of = [cap](const void* symbol, char* out) {
        if (!out || cap <= 0) return;
        Dl_info info;
        const char* name = (dladdr(symbol, &info) != 0 && info.dli_fname) ? info.dli_fname : "";
        snprintf(out, (size_t)cap, "%s", name);
    };
    object_of(reinterpret_cast<const void*>(&hipStreamSynchronize), hip_path);
    object_of(reinterpret_cast<const void*>(&ncclAllReduce), rccl_path);
    if (hip_runtime_version) {
        int v = 0;
        if (hipRuntimeGetVersion(&v) != hipSuccess) v = -1;      // (no device needed: the runtime's own build number)
        *hip_runtime_version = v;
    }
    if (rccl_version) {
        int v = 0;
        if (ncclGetVersion(&v) != ncclSuccess) v = -1;
        *rccl_version = v;
    }
    return MI_OK;
}

extern "C" int mi_shard_range(int m_total, int rank, int world, int* lo, int* hi)
{
    if (m_total < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) { set_error("mi_shard_range: bad arguments"); return MI_ERR_INVALID_ARG; }
    *lo = (int)((long long)m_total * rank / world);
    *hi = (int)((long long)m_total * (rank + 1) / world);
    return MI_OK;
}

extern "C" int mi_source_share(int n_total, int rank, int world, int* count)
{
    if (n_total < 0 || world < 1 || rank < 0 || rank >= world || !count) { set_error("mi_source_share: bad arguments"); return MI_ERR_INVALID_ARG; }
    if (n_total < 4 * ICP_CHUNK_POINTS * world) {                                   // too few chunks to deal: contiguous slices
        *count = (int)((long long)n_total * (rank + 1) / world) - (int)((long long)n_total * rank / world);
        return MI_OK;
    }
    const int chunks = (n_total + ICP_CHUNK_POINTS - 1) / ICP_CHUNK_POINTS;
    const int mine = (chunks - rank + world - 1) / world;                           // chunks rank, rank + W, ...
    const bool has_last = (chunks - 1) % world == rank;                             // the (possibly partial) last chunk of the cloud
    *count = mine * ICP_CHUNK_POINTS - (has_last ? chunks * ICP_CHUNK_POINTS - n_total : 0);
    return MI_OK;
}

extern "C" unsigned long long mi_pack_key(float d2, int global_index)
{
    unsigned int bits;
    memcpy(&bits, &d2, sizeof bits);
    return ((unsigned long long)bits << 32) | (unsigned int)global_index;
}

extern "C" void mi_unpack_key(unsigned long long key, float* d2, int* global_index)
{
    const unsigned int bits = (unsigned int)(key >> 32);
    if (d2) memcpy(d2, &bits, sizeof bits);
    if (global_index) *global_index = (int)(unsigned int)(key & 0xffffffffull);
}

namespace mislam { void cpd_workspace_destroy(mi_ctx* ctx); }

extern "C" void mi_ctx_destroy(mi_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->aux) (void)hipStreamSynchronize(c->aux);
    if (c->aux2) (void)hipStreamSynchronize(c->aux2);
    retire_buffers(c);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->exchange_host) (void)hipHostFree(c->exchange_host);
    for (hipEvent_t e : c->pin_event) if (e) (void)hipEventDestroy(e);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->h_scratch) (void)hipHostFree(c->h_scratch);
    cpd_workspace_destroy(c);
    c->staging.release(); c->staging2.release(); c->tcodes2_in.release(); c->tcodes2_out.release(); c->torder2_in.release(); c->tbbox2.release();
    c->gbbox.release(); c->tsort_temp2.release();
    c->bx.release(); c->by.release(); c->bz.release();
    c->cx.release(); c->cy.release(); c->cz.release(); c->ax.release(); c->ay.release(); c->az.release();
    c->tx.release(); c->ty.release(); c->tz.release();
    c->tgt4.release(); c->keys.release(); c->part_mom.release(); c->part_err.release();
    c->idx_tmp.release(); c->keep_tmp.release();
    c->tcodes_in.release(); c->tcodes_out.release(); c->torder_in.release(); c->torder_out.release();
    c->tbbox.release(); c->tsort_temp.release(); c->tpts.release(); c->tboxes.release(); c->sorder.release(); c->sinv.release(); c->resid.release();
    c->tleaf.release(); c->tidx.release(); c->tboxes6.release(); c->nn_stats.release();
    c->gpts.release(); c->gstart.release(); c->gfill.release(); c->gscan.release(); c->rows.release(); c->rows_reduced.release();
    c->sched_order.release(); c->sched_far.release(); c->sched_lanes.release(); c->sched_counters.release(); c->gslot_of.release(); c->match_slot.release();
    c->grow_occ.release(); c->gnear_tmp.release();
    for (auto& s : c->spans) { (void)hipEventDestroy(s.e0); (void)hipEventDestroy(s.e1); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->d_state) (void)hipFree(c->d_state);
    if (c->h_state) (void)hipHostFree(c->h_state);
    if (c->peek_event) (void)hipEventDestroy(c->peek_event);
    for (hipEvent_t e : c->aux_event) if (e) (void)hipEventDestroy(e);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->aux2) (void)hipStreamDestroy(c->aux2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    pool_context_gone(c->device);
    delete c;
}

extern "C" int mi_ctx_synchronize(mi_ctx* c)
{
    if (!c) { set_error("mi_ctx_synchronize: null context"); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    MI_HIP(hipStreamSynchronize(c->stream));
    retire_buffers(c);
    return MI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// profiling: HIP events on the context's own stream around each kernel launch
// ---------------------------------------------------------------------------------------------------------------
// Timing events WITHOUT the system-scope release a default event performs when it is recorded (hipEventDisableSystemFence): behind the search
// kernel that release writes back ~20 MB of dirty L2 lines, twice per step.  MISLAM_PROF_EVENT_FLAGS overrides (developer switch).
static unsigned int c_prof_event_flags()
{
    static const unsigned int flags = [] { const char* e = getenv("MISLAM_PROF_EVENT_FLAGS"); return e ? (unsigned int)strtoul(e, nullptr, 0) : (unsigned int)hipEventDisableSystemFence; }();
    return flags;
}

int mi_ctx::prof_begin(int kernel)
{
    ProfileSpan s{};
    s.kernel = kernel;
    for (hipEvent_t* e : {&s.e0, &s.e1}) {
        if (!event_pool.empty()) { *e = event_pool.back(); event_pool.pop_back(); }
        else MI_HIP(hipEventCreateWithFlags(e, c_prof_event_flags()));
    }
    MI_HIP(hipEventRecord(s.e0, stream));
    spans.push_back(s);
    return MI_OK;
}

int mi_ctx::prof_span(int kernel, hipEvent_t* e0, hipEvent_t* e1)
{
    *e0 = *e1 = nullptr;
    if (!(profile && ((prof_mask >> kernel) & 1u))) return MI_OK;
    ProfileSpan s{};
    s.kernel = kernel;
    for (hipEvent_t* e : {&s.e0, &s.e1}) {
        if (!event_pool.empty()) { *e = event_pool.back(); event_pool.pop_back(); }
        else MI_HIP(hipEventCreateWithFlags(e, c_prof_event_flags()));
    }
    spans.push_back(s);
    *e0 = s.e0; *e1 = s.e1;
    return MI_OK;
}

int mi_ctx::prof_end()
{
    MI_HIP(hipEventRecord(spans.back().e1, stream));
    return MI_OK;
}

int mi_ctx::prof_collect()
{
    if (spans.empty()) return MI_OK;
    MI_HIP(hipStreamSynchronize(stream));
    for (auto& s : spans) {
        float ms = 0.f;
        MI_HIP(hipEventElapsedTime(&ms, s.e0, s.e1));
        prof_ms[s.kernel] += ms;
        prof_n[s.kernel] += 1;
        event_pool.push_back(s.e0);
        event_pool.push_back(s.e1);
    }
    spans.clear();
    return MI_OK;
}

extern "C" int mi_profile_enable(mi_ctx* c, int enable)
{
    if (!c) { set_error("mi_profile_enable: null context"); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    MI_TRY(c->prof_collect());
    c->profile = enable != 0;
    return MI_OK;
}

extern "C" int mi_profile_select(mi_ctx* c, unsigned int kernel_mask)
{
    if (!c) { set_error("mi_profile_select: null context"); return MI_ERR_INVALID_ARG; }
    c->prof_mask = kernel_mask;
    return MI_OK;
}

extern "C" int mi_profile_reset(mi_ctx* c)
{
    if (!c) { set_error("mi_profile_reset: null context"); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    MI_TRY(c->prof_collect());
    for (int k = 0; k < MI_KERNEL_COUNT; k++) { c->prof_ms[k] = 0; c->prof_n[k] = 0; }
    return MI_OK;
}

// The counting build's counters (GRID_STATS_ROWS copies of GRID_STATS_COLS words), summed over the copies ([7] is a maximum)
static int search_counters(mi_ctx* c, unsigned long long (&sum)[GRID_STATS_COLS])
{
    const size_t words = (size_t)GRID_STATS_ROWS * GRID_STATS_COLS;
    for (int i = 0; i < GRID_STATS_COLS; i++) sum[i] = 0;
    if (!c->nn_stats_on) return MI_OK;
    std::vector<unsigned long long> h(words);
    MI_HIP(hipMemcpyAsync(h.data(), c->nn_stats.p, sizeof(unsigned long long) * words, hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipStreamSynchronize(c->stream));
    for (size_t r = 0; r < (size_t)GRID_STATS_ROWS; r++)
        for (int i = 0; i < GRID_STATS_COLS; i++) sum[i] = i == 7 ? std::max(sum[i], h[r * GRID_STATS_COLS + i]) : sum[i] + h[r * GRID_STATS_COLS + i];
    return MI_OK;
}

extern "C" int mi_profile_search_stats(mi_ctx* c, int enable, unsigned long long out[8])
{
    if (!c) { set_error("mi_profile_search_stats: null context"); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    const size_t words = (size_t)GRID_STATS_ROWS * GRID_STATS_COLS;
#ifdef MISLAM_DEV_WAVE_TIMELINE        // developer build: 4 words per wave of the last search behind the counters, dumped to $MISLAM_DEV_TIMELINE_FILE
    const size_t tl_words = (size_t)16 * (1u << 18);
    MI_TRY(c->nn_stats.reserve(words + tl_words));
    if (out && c->nn_stats_on && getenv("MISLAM_DEV_TIMELINE_FILE")) {
        std::vector<unsigned long long> tl(tl_words);
        MI_HIP(hipMemcpyAsync(tl.data(), c->nn_stats.p + words, sizeof(unsigned long long) * tl_words, hipMemcpyDeviceToHost, c->stream));
        MI_HIP(hipStreamSynchronize(c->stream));
        if (FILE* f = fopen(getenv("MISLAM_DEV_TIMELINE_FILE"), "wb")) { fwrite(tl.data(), sizeof(unsigned long long), tl_words, f); fclose(f); }
    }
#endif
    MI_TRY(c->nn_stats.reserve(words));
    if (out) {
        unsigned long long sum[GRID_STATS_COLS];
        MI_TRY(search_counters(c, sum));
        for (int i = 0; i < 8; i++) out[i] = sum[i];
    }
    c->nn_stats_on = enable != 0;
    if (enable) MI_HIP(hipMemsetAsync(c->nn_stats.p, 0, sizeof(unsigned long long) * words, c->stream));
    return MI_OK;
}

extern "C" int mi_profile_search_phases(mi_ctx* c, unsigned long long out[20])
{
    if (!c || !out) { set_error("mi_profile_search_phases: null argument"); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    if (!c->nn_stats_on) { set_error("mi_profile_search_phases: counting is off (mi_profile_search_stats(ctx, 1, NULL) first)"); return MI_ERR_STATE; }
    unsigned long long sum[GRID_STATS_COLS];
    MI_TRY(search_counters(c, sum));
    static_assert(GRID_STATS_COLS >= 8 + 20, "phases: 20 counters behind the 8 of mi_profile_search_stats");
    for (int i = 0; i < 20; i++) out[i] = sum[8 + i];
    return MI_OK;
}

extern "C" int mi_selftest_fail_loads(mi_ctx* c, int n)
{
    if (!c || n < 0) { set_error("mi_selftest_fail_loads: bad argument"); return MI_ERR_INVALID_ARG; }
    c->selftest_fail_loads = n;
    return MI_OK;
}

extern "C" int mi_selftest_sort_pairs(mi_ctx* c, unsigned int* keys, int* values, int n, int bits)
{
    if (!c || n < 0 || (n > 0 && (!keys || !values)) || (bits != 10 && bits != 20 && bits != 30)) {
        set_error("mi_selftest_sort_pairs: bad argument");
        return MI_ERR_INVALID_ARG;
    }
    if (n == 0) return MI_OK;
    MI_ENTER(c);
    DevBuf<unsigned int> k0, k1;
    DevBuf<int> v0, v1;
    DevBuf<unsigned char> temp;
    MI_TRY(k0.reserve((size_t)n)); MI_TRY(k1.reserve((size_t)n)); MI_TRY(v0.reserve((size_t)n)); MI_TRY(v1.reserve((size_t)n));
    MI_TRY(temp.reserve(radix_sort_temp_bytes(n)));
    MI_HIP(hipMemcpyAsync(k0.p, keys, sizeof(unsigned int) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    MI_HIP(hipMemcpyAsync(v0.p, values, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    MI_HIP(radix_sort_pairs_u32(temp.p, k0.p, k1.p, v0.p, v1.p, n, bits, c->stream));
    MI_HIP(hipMemcpyAsync(keys, k1.p, sizeof(unsigned int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipMemcpyAsync(values, v1.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipStreamSynchronize(c->stream));
    k0.release(); k1.release(); v0.release(); v1.release(); temp.release();      // (DevBuf has no destructor: the call's own scratch goes back to the pool here)
    return MI_OK;
}

extern "C" int mi_profile_get(mi_ctx* c, int kernel, double* total_ms, long long* launches)
{
    if (!c || kernel < 0 || kernel >= MI_KERNEL_COUNT) { set_error("mi_profile_get: bad argument"); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    MI_TRY(c->prof_collect());
    if (total_ms) *total_ms = c->prof_ms[kernel];
    if (launches) *launches = c->prof_n[kernel];
    return MI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// shared helpers
// ---------------------------------------------------------------------------------------------------------------
namespace mislam {

static inline int round_up(int v, int g) { return (v + g - 1) / g * g; }

constexpr int NN_MAX_CHUNKS = 1024;

size_t target_alloc_len(int m_local)
{
    // any chunking with <= NN_MAX_CHUNKS chunks of T-aligned length stays inside this allocation
    return (size_t)round_up(std::max(m_local, 1), NN_TARGET_BLOCK) + (size_t)NN_TARGET_BLOCK * NN_MAX_CHUNKS;
}

// 2-D decomposition of the (source, target) pair space for K1.  Few chunks = few re-scan restarts and few atomics;
// enough workgroups = every CU busy with a short tail.  Measured on MI355X (profiles/r01_nn_microbench.log): R = 2 with
// the smallest chunk count that still yields >= ~8 workgroups per CU is the fastest configuration at every size.
NnPlan plan_nn(const mi_ctx* ctx, int n, int m_local)
{
    NnPlan p;
    p.R = ctx->tune.nn_R;
    if (p.R != 1 && p.R != 2 && p.R != 4 && p.R != 8) p.R = 2;
    const int n_src_blocks = round_up(std::max(n, 1), 256 * p.R) / (256 * p.R);
    const int target_wgs = ctx->tune.nn_wgs > 0 ? ctx->tune.nn_wgs : ctx->cu_count * 8;
    int chunks = (target_wgs + n_src_blocks - 1) / n_src_blocks;
    // a chunk should fit an XCD's L2 next to everything else it holds: <= 2 MB of target xyz (12 B/point)
    const int l2_chunks = (int)(((long long)std::max(m_local, 1) * 12 + (2 << 20) - 1) / (2 << 20));
    chunks = std::max(chunks, l2_chunks);
    const int max_chunks = std::max(1, std::min(NN_MAX_CHUNKS, m_local / (NN_TARGET_BLOCK * 4)));
    chunks = std::max(1, std::min(chunks, max_chunks));
    // multiples of 8 get the XCD-pinned block mapping of K1
    if (chunks > 1 && max_chunks >= 8) chunks = std::min(round_up(chunks, 8), max_chunks / 8 * 8);
    const int forced = ctx->tune.nn_chunks;
    if (forced > 0) chunks = std::min(forced, NN_MAX_CHUNKS);
    p.chunk_len = round_up((std::max(m_local, 1) + chunks - 1) / chunks, NN_TARGET_BLOCK);
    p.n_chunks = (std::max(m_local, 1) + p.chunk_len - 1) / p.chunk_len;
    return p;
}

int host_to_device(mi_ctx* c, void* dst_dev, const void* src_host, size_t bytes)
{
    const hipStream_t ws = c->work_stream();
    constexpr size_t PIECE = mi_ctx::PIN_PIECE;
    if (bytes < PIECE / 4 || c->pin == nullptr) {     // small: the runtime's path is fine (and synchronous for pageable memory)
        MI_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ws));
        return MI_OK;
    }
    for (size_t o = 0; o < bytes; o += PIECE) {       // the copy of piece k overlaps the transfers of the pieces before it
        const size_t nb = std::min(PIECE, bytes - o);
        const unsigned int k = c->pin_next++ % mi_ctx::PIN_SLOTS;
        if (c->pin_busy & (1u << k)) { MI_HIP(hipEventSynchronize(c->pin_event[k])); c->pin_busy &= ~(1u << k); }   // (sixteen pieces ago: long done)
        char* slot = c->pin + (size_t)k * PIECE;
        memcpy(slot, (const char*)src_host + o, nb);
        MI_HIP(hipMemcpyAsync((char*)dst_dev + o, slot, nb, hipMemcpyHostToDevice, ws));
        MI_HIP(hipEventRecord(c->pin_event[k], ws));
        c->pin_busy |= 1u << k;
    }
    return MI_OK;
}

int upload_soa(mi_ctx* c, const float* host_aos, int n, int n_pad, float* x, float* y, float* z, float4* packed)
{
    DevBuf<float>& staging = c->lane == 1 ? c->staging2 : c->staging;
    MI_TRY(staging.reserve((size_t)3 * n));
    MI_TRY(host_to_device(c, staging.p, host_aos, sizeof(float) * 3 * (size_t)n));
    MI_HIP(aos_to_soa(staging.p, n, n_pad, x, y, z, packed, c->work_stream()));
    // the staging buffer is reused by the next upload of the same lane: stream order keeps them apart, and a pageable-memory copy is
    // already synchronous with respect to the host buffer
    return MI_OK;
}

// Scratch for one Morton sort of m points (shared by the fixed-cloud hierarchy and the moving-cloud ordering).
static int morton_args(mi_ctx* c, const float* x, const float* y, const float* z, int m, int* order_out, MortonArgs* out)
{
    const size_t sort_bytes = tree_sort_temp_bytes(m);
    const bool second = c->lane == 1;                   // (the lane's own scratch set: two sorts may be in flight, one per lane)
    DevBuf<unsigned int>& codes_in = second ? c->tcodes2_in : c->tcodes_in;
    DevBuf<unsigned int>& codes_out = second ? c->tcodes2_out : c->tcodes_out;
    DevBuf<int>& order_in = second ? c->torder2_in : c->torder_in;
    DevBuf<float>& bbox = second ? c->tbbox2 : c->tbbox;
    DevBuf<unsigned char>& temp = second ? c->tsort_temp2 : c->tsort_temp;
    MI_TRY(codes_in.reserve((size_t)m)); MI_TRY(codes_out.reserve((size_t)m));
    MI_TRY(order_in.reserve((size_t)m));
    MI_TRY(bbox.reserve(256 * 6 + 8));
    MI_TRY(temp.reserve(sort_bytes + 16));
    MortonArgs a{};
    a.x = x; a.y = y; a.z = z; a.m = m;
    a.bbox_partials = bbox.p; a.bbox = bbox.p + 256 * 6;
    a.codes_in = codes_in.p; a.codes_out = codes_out.p; a.order_in = order_in.p; a.order_out = order_out;
    a.sort_temp = temp.p; a.sort_temp_bytes = sort_bytes;
    *out = a;
    return MI_OK;
}

// Builds the box hierarchy over the resident fixed-cloud shard if it is not there yet (once per mi_icp_load / search).
static int ensure_tree(mi_ctx* c, int m_local, int index_base)
{
    if (c->tree_valid) return MI_OK;
    const int n_leaves = (m_local + TREE_LEAF - 1) / TREE_LEAF;
    int n_pad = 1, height = 0;
    while (n_pad < n_leaves) { n_pad <<= 1; height++; }
    if (height > TREE_MAX_HEIGHT) { set_error("fixed cloud too large for the box hierarchy"); return MI_ERR_INVALID_ARG; }
    if (c->selftest_fail_loads > 0) {      // mi_selftest_fail_loads (tests): the next N index builds of the context fail HERE -- behind the fixed cloud's
        c->selftest_fail_loads -= 1;       // upload, which is already on the auxiliary stream: the early return mi_icp_load's lane guard exists for
        set_error("index build failed on request (mi_selftest_fail_loads)");
        return MI_ERR_INVALID_ARG;
    }
    MI_TRY(c->torder_out.reserve((size_t)m_local));
    MI_TRY(c->tpts.reserve((size_t)n_leaves * TREE_LEAF));
    MI_TRY(c->tboxes.reserve((size_t)4 * n_pad));
    MI_TRY(c->tleaf.reserve((size_t)n_leaves * (3 * TREE_LEAF / 4)));
    MI_TRY(c->tidx.reserve((size_t)n_leaves * TREE_LEAF));
    MI_TRY(c->tboxes6.reserve((size_t)12 * ((size_t)n_pad + 6)));        // pairs of nodes (nn_tree.h), incl. the padding a step may read
    TreeBuildArgs a{};
    MI_TRY(morton_args(c, c->tx.p, c->ty.p, c->tz.p, m_local, c->torder_out.p, &a.morton));
    a.index_base = index_base; a.n_leaves = n_leaves; a.n_pad = n_pad;
    a.pts = c->tpts.p; a.boxes = c->tboxes.p;
    a.leaf_soa = c->tleaf.p; a.leaf_idx = c->tidx.p; a.boxes6 = c->tboxes6.p;
    MI_HIP(tree_build(a, c->work_stream()));
    c->tree.boxes6 = c->tboxes6.p;
    c->tree.leaf_soa = c->tleaf.p; c->tree.leaf_idx = c->tidx.p;
    c->tree.n_pad = n_pad; c->tree.height = height; c->tree.n_leaves = n_leaves;
    c->tree_valid = true;
    return MI_OK;
}

// Builds the cell grid over the resident fixed-cloud shard if it is not there yet.  The cell size comes from the cloud's bounding
// box, which the host reads back: one stream synchronisation per fixed cloud, at load time.
static int ensure_grid(mi_ctx* c, int m_local, int index_base)
{
    if (c->grid_valid) return MI_OK;
    const hipStream_t ws = c->work_stream();
    MI_TRY(c->gbbox.reserve(256 * 6 + 8));              // (its own: a Morton sort's bounding box may be in flight on another lane)
    float* d_bbox = c->gbbox.p + 256 * 6;
    MI_HIP(cloud_bbox(c->tx.p, c->ty.p, c->tz.p, m_local, c->gbbox.p, d_bbox, ws));
    float* bbox = c->h_scratch;                        // (pinned: a read-back into pageable memory goes through the runtime's staging)
    MI_HIP(hipMemcpyAsync(bbox, d_bbox, 6 * sizeof(float), hipMemcpyDeviceToHost, ws));
    { StallProbe sp("grid: bounding-box synchronize"); MI_HIP(hipStreamSynchronize(ws)); }
    StallProbe sp_rest("grid: reserve + enqueue build");
    NnGridView g{};
    grid_plan(bbox, m_local, c->tune.grid_points_per_cell, &g);
    const size_t n_cells = (size_t)g.nx * g.ny * g.nz;
    MI_TRY(c->gpts.reserve((size_t)m_local + GRID_PTS_PAD));
    MI_TRY(c->grow_occ.reserve(n_cells)); MI_TRY(c->gnear_tmp.reserve(n_cells));
    MI_TRY(c->gstart.reserve(n_cells + 1 + 3));           // (+3: a row's offsets are fetched four words at a time)
    MI_TRY(c->gfill.reserve(n_cells + 1));
    MI_TRY(c->gscan.reserve((n_cells + 1) / 1024 + 2));
    MI_TRY(c->gslot_of.reserve((size_t)m_local));
    g.pts = c->gpts.p;
    g.cell_start = c->gstart.p;
    g.slot_of = c->gslot_of.p;
    g.row_occ = c->grow_occ.p;
    g.index_base = index_base;
    GridBuildArgs a{};
    a.x = c->tx.p; a.y = c->ty.p; a.z = c->tz.p; a.m = m_local; a.index_base = index_base;
    a.view = g; a.cell_fill = c->gfill.p; a.scan_tmp = c->gscan.p; a.pts_out = c->gpts.p; a.cell_start_out = c->gstart.p; a.slot_of_out = c->gslot_of.p;
    a.row_occ_out = c->grow_occ.p; a.near_tmp = c->gnear_tmp.p;
    MI_HIP(grid_build(a, ws));
    c->grid = g;
    c->grid_valid = true;
    return MI_OK;
}

// Morton-sorts the moving cloud once: src (SoA, n real points) -> dst (SoA, n_pad entries, tail = copies of the last sorted
// point); c->sorder[s] = the caller's index of sorted slot s.  Spatially adjacent sources then share a wave, which is what
// makes the wave-cooperative hierarchy walk tight; K2-K6 are order-agnostic sums, so nothing else changes.
static int sort_sources(mi_ctx* c, const float* sx, const float* sy, const float* sz, int n, int n_pad, float* dx, float* dy, float* dz)
{
    MI_TRY(c->sorder.reserve((size_t)n));
    MortonArgs ma{};
    MI_TRY(morton_args(c, sx, sy, sz, n, c->sorder.p, &ma));
    MI_HIP(morton_order(ma, c->work_stream()));
    MI_HIP(permute_soa(sx, sy, sz, c->sorder.p, n, n_pad, dx, dy, dz, c->work_stream()));
    return MI_OK;
}

int resolve_nn_mode(const mi_ctx* c, int nn_mode, int m_local)
{
    const int forced = c->tune.nn_force_mode;
    if (forced == MI_NN_BRUTEFORCE || forced == MI_NN_TREE || forced == MI_NN_GRID) nn_mode = forced;
    if (nn_mode == MI_NN_BRUTEFORCE || nn_mode == MI_NN_TREE || nn_mode == MI_NN_GRID) return nn_mode;
    // measured crossover on MI355X, every-pair against the cell grid, ms per ICP step at N = M (profiles/r03_crossover.log):
    // 6 000: 0.042 / 0.042, 8 000: 0.048 / 0.046, 10 000: 0.056 / 0.048, 12 000: 0.063 / 0.047, 16 000: 0.085 / 0.051 -- the grid's
    // index builds (0.3 ms per registration) are what keeps the switch at 10 000 rather than 7 000
    return m_local >= MI_NN_INDEX_MIN_POINTS ? MI_NN_GRID : MI_NN_BRUTEFORCE;
}

extern "C" const char* mi_nn_kernel_name(const mi_ctx* c, int n_moving, int m_fixed_local, int nn_mode)
{
    (void)n_moving;
    if (!c) return "";
    const int mode = resolve_nn_mode(c, nn_mode, m_fixed_local);
    return mode == MI_NN_GRID ? nn_grid_kernel_name(false) : (mode == MI_NN_TREE ? "nn_tree_kernel" : "nn_bruteforce_kernel");
}

int launch_nn(mi_ctx* c, const float* sx, const float* sy, const float* sz, int n, int m_local, int index_base, int fma,
                      const int* done_flag, int nn_mode)
{
    const int mode = resolve_nn_mode(c, nn_mode, m_local);
    if (mode == MI_NN_TREE || mode == MI_NN_GRID) {
        MI_TRY(ensure_tree(c, m_local, index_base));
        if (mode == MI_NN_GRID) MI_TRY(ensure_grid(c, m_local, index_base));
        ProfScope ps(c, MI_KERNEL_NN);
        if (mode == MI_NN_GRID) {
            GridSearchArgs a{};
            a.sx = sx; a.sy = sy; a.sz = sz; a.done_flag = done_flag; a.n = n; a.keys = c->keys.p;
            a.stats = c->nn_stats_on ? c->nn_stats.p : nullptr;
            a.deal_rows = c->tune.grid_deal_rows < 0 ? (n >= GRID_DEAL_ROWS_MIN_POINTS ? 1 : 0) : c->tune.grid_deal_rows;
            MI_HIP(nn_grid_query(c->grid, c->tree, a, fma, c->stream));
        } else {
            MI_HIP(nn_tree_query(c->tree, sx, sy, sz, n, c->keys.p, done_flag, fma, c->stream));
        }
        return MI_OK;
    }
    const NnPlan p = plan_nn(c, n, m_local);
    NnLaunch a{};
    a.sx = sx; a.sy = sy; a.sz = sz;
    a.n = n; a.n_pad = round_up(n, 256 * p.R);
    a.tx = c->tx.p; a.ty = c->ty.p; a.tz = c->tz.p;
    a.chunk_len = p.chunk_len; a.n_chunks = p.n_chunks;
    a.index_base = index_base;
    a.keys = c->keys.p;
    a.done_flag = done_flag;
    a.R = p.R;
    a.fma = fma;
    // host-side shape checks before a hand-written kernel runs (a fault can reset the whole node)
    if ((size_t)a.n_pad > c->cx.cap && sx == c->cx.p) { set_error("internal: source padding exceeds allocation"); return MI_ERR_STATE; }
    if ((size_t)a.n_chunks * a.chunk_len > c->tx.cap) { set_error("internal: target chunking exceeds allocation"); return MI_ERR_STATE; }
    if (a.chunk_len % NN_TARGET_BLOCK != 0) { set_error("internal: chunk_len not a multiple of the target block"); return MI_ERR_STATE; }
    ProfScope ps(c, MI_KERNEL_NN);
    MI_HIP(nn_launch(a, c->stream));
    return MI_OK;
}

static int allreduce_keys(mi_ctx* c, int n)
{
    if (!c->distributed()) return MI_OK;
    ProfScope ps(c, MI_KERNEL_ALLREDUCE);
    return allreduce_min_u64(c, c->keys.p, (size_t)n);
}

static int allreduce_doubles(mi_ctx* c, double* dev_ptr, int count) { return allreduce_sum_f64(c, dev_ptr, (size_t)count); }

static void shard_range(int m_total, int rank, int world, int* lo, int* hi) { (void)mi_shard_range(m_total, rank, world, lo, hi); }

// Uploads this rank's shard of the fixed cloud (SoA streams for K1 + float4 for gathers).
int upload_target_shard(mi_ctx* c, const float* after_xyz, int m_total, bool replicate)
{
    c->m_total = m_total;
    c->tree_valid = false;   // the indexes cover the previous shard
    c->grid_valid = false;
    if (replicate) { c->shard_lo = 0; c->shard_hi = m_total; }      // source-sharded: every rank holds the whole fixed cloud
    else shard_range(m_total, c->rank, c->world, &c->shard_lo, &c->shard_hi);
    const int m_local = c->shard_hi - c->shard_lo;
    const size_t len = target_alloc_len(m_local);
    MI_TRY(c->tx.reserve(len)); MI_TRY(c->ty.reserve(len)); MI_TRY(c->tz.reserve(len));
    MI_TRY(c->tgt4.reserve(len));
    if (m_local > 0)
        MI_TRY(upload_soa(c, after_xyz + 3 * (size_t)c->shard_lo, m_local, (int)len, c->tx.p, c->ty.p, c->tz.p, c->tgt4.p));
    return MI_OK;
}

// rows of partial sums for the loaded moving cloud (icp_rows.hpp) + the reduced rows
static int reserve_rows(mi_ctx* c)
{
    const size_t rows = (size_t)icp_row_count(c->n_pad) * (ICP_MOMENTS + ICP_ERRSUMS);
    if (rows > c->rows.cap) {
        MI_TRY(c->rows.reserve(rows));
        MI_HIP(hipMemsetAsync(c->rows.p, 0, sizeof(double) * c->rows.cap, c->stream));   // columns a path never writes stay finite
    }
    MI_TRY(c->rows_reduced.reserve((size_t)ICP_REDUCED_ROWS * (ICP_MOMENTS + ICP_ERRSUMS)));
    MI_HIP(hipMemsetAsync(c->rows_reduced.p, 0, sizeof(double) * ICP_REDUCED_ROWS * (ICP_MOMENTS + ICP_ERRSUMS), c->stream));   // (rows past a rank's count: zero)
    MI_TRY(c->sched_order.reserve((size_t)icp_row_count(c->n_pad)));
    MI_TRY(c->sched_far.reserve((size_t)icp_row_count(c->n_pad)));
    MI_TRY(c->sched_lanes.reserve((size_t)icp_row_count(c->n_pad)));
    MI_TRY(c->sched_counters.reserve(2));
    return MI_OK;
}

static IcpSchedule make_schedule(mi_ctx* c)
{
    IcpSchedule s{};
    s.order = c->sched_order.p; s.far = c->sched_far.p; s.counters = c->sched_counters.p; s.lanes = c->sched_lanes.p;
    return s;
}

static IcpView make_view(mi_ctx* c)
{
    IcpView v{};
    v.state = c->d_state;
    v.bx = c->bx.p; v.by = c->by.p; v.bz = c->bz.p;
    v.cx = c->cx.p; v.cy = c->cy.p; v.cz = c->cz.p;
    v.tgt4 = c->tgt4.p;
    v.keys = c->keys.p;
    v.n = c->n; v.n_pad = c->n_pad;
    v.shard_lo = c->shard_lo; v.shard_hi = c->shard_hi;
    v.filter_pairs = c->icp.filter_pairs;
    v.max_distance_squared = c->icp.max_distance_squared;
    v.fma = c->icp.dist_mode == MI_DIST_FMA;
    const bool seq = c->icp.sum_mode == MI_SUM_CPU_SEQUENTIAL;
    v.inv_order = seq ? c->sinv.p : nullptr;
    v.resid = seq ? c->resid.p : nullptr;
    return v;
}

static void state_identity(IcpState* s)
{
    memset(s, 0, sizeof *s);
    s->R[0] = s->R[4] = s->R[8] = 1.f;
    s->prevR[0] = s->prevR[4] = s->prevR[8] = 1.f;
    s->error = 1e5f;                 // "*error = 1e5"  basicicp.cpp:26
    s->prev_error = 3.402823466e38f; // numeric_limits<float>::max()  icpcuda.cu:10
}

}  // namespace mislam

// ---------------------------------------------------------------------------------------------------------------
// ICP
// ---------------------------------------------------------------------------------------------------------------
extern "C" void mi_icp_params_default(mi_icp_params* p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->eps = 1e-3f;
    p->max_iterations = -1;
    p->max_distance_squared = 1000.f;
    p->dist_mode = MI_DIST_CPU_ROUNDING;
    p->compose_mode = MI_COMPOSE_CPU_ADDITIVE;
    p->filter_pairs = 1;
    p->abort_on_increase = 0;
    p->sync_every = 0;
    p->verbose = 0;
}

extern "C" void mi_icp_params_cuda_slam(mi_icp_params* p)
{
    if (!p) return;
    mi_icp_params_default(p);
    p->dist_mode = MI_DIST_FMA;
    p->compose_mode = MI_COMPOSE_EXACT;
    p->filter_pairs = 0;
    p->abort_on_increase = 1;
}

static int icp_check_params(const mi_icp_params* p)
{
    if (!p) { set_error("ICP: null params"); return MI_ERR_INVALID_ARG; }
    if (p->dist_mode != MI_DIST_CPU_ROUNDING && p->dist_mode != MI_DIST_FMA) { set_error("ICP: bad dist_mode %d", p->dist_mode); return MI_ERR_INVALID_ARG; }
    if (p->compose_mode != MI_COMPOSE_CPU_ADDITIVE && p->compose_mode != MI_COMPOSE_EXACT) { set_error("ICP: bad compose_mode %d", p->compose_mode); return MI_ERR_INVALID_ARG; }
    if (p->max_iterations < -1) { set_error("ICP: max_iterations %d (use -1 for unbounded)", p->max_iterations); return MI_ERR_INVALID_ARG; }
    if (p->nn_mode != MI_NN_AUTO && p->nn_mode != MI_NN_BRUTEFORCE && p->nn_mode != MI_NN_TREE && p->nn_mode != MI_NN_GRID) { set_error("ICP: bad nn_mode %d", p->nn_mode); return MI_ERR_INVALID_ARG; }
    if (p->shard_mode != MI_SHARD_AUTO && p->shard_mode != MI_SHARD_TARGET && p->shard_mode != MI_SHARD_SOURCE) { set_error("ICP: bad shard_mode %d", p->shard_mode); return MI_ERR_INVALID_ARG; }
    if (p->sum_mode != MI_SUM_EXACT && p->sum_mode != MI_SUM_CPU_SEQUENTIAL) { set_error("ICP: bad sum_mode %d", p->sum_mode); return MI_ERR_INVALID_ARG; }
    return MI_OK;
}

extern "C" int mi_icp_reset(mi_ctx* c)
{
    if (!c || !c->icp_loaded) { set_error("mi_icp_reset: no problem loaded"); return MI_ERR_STATE; }
    MI_ENTER(c);
    state_identity(c->h_state);
    c->enqueued_passes = 0;
    if (c->icp.max_iterations == 0) {   // "while (iterations < maxIterations)" never enters
        c->h_state->done = 1;
        c->h_state->stop_reason = MI_STOP_MAX_ITERATIONS;
    }
    MI_HIP(hipMemcpyAsync(c->d_state, c->h_state, sizeof(IcpState), hipMemcpyHostToDevice, c->stream));
    const size_t bytes = sizeof(float) * (size_t)c->n_pad;
    MI_HIP(hipMemcpyAsync(c->cx.p, c->bx.p, bytes, hipMemcpyDeviceToDevice, c->stream));   // transformedCloud = cloudBefore, basicicp.cpp:30
    MI_HIP(hipMemcpyAsync(c->cy.p, c->by.p, bytes, hipMemcpyDeviceToDevice, c->stream));
    MI_HIP(hipMemcpyAsync(c->cz.p, c->bz.p, bytes, hipMemcpyDeviceToDevice, c->stream));
    MI_HIP(fill_keys(c->keys.p, c->n, c->stream));
    if (c->fused) {
        MI_TRY(c->match_slot.reserve((size_t)c->n_pad));
        MI_HIP(hipMemsetAsync(c->match_slot.p, 0xff, sizeof(unsigned int) * (size_t)c->n, c->stream));   // ~0u: no match yet
    }
    MI_HIP(icp_schedule_reset(make_schedule(c), icp_row_count(c->n), c->stream));
    MI_HIP(hipStreamSynchronize(c->stream));
    retire_buffers(c);
    return MI_OK;
}

extern "C" int mi_icp_load(mi_ctx* c, const float* before_xyz, int n_before, const float* after_xyz, int n_after,
                           const mi_icp_params* params)
{
    if (!c) { set_error("mi_icp_load: null context"); return MI_ERR_INVALID_ARG; }
    if (!before_xyz || !after_xyz || n_before <= 0 || n_after <= 0) { set_error("mi_icp_load: empty or null cloud (n_before=%d, n_after=%d)", n_before, n_after); return MI_ERR_INVALID_ARG; }
    if (n_after < c->world || n_before < c->world) { set_error("mi_icp_load: fewer points (%d, %d) than ranks (%d)", n_before, n_after, c->world); return MI_ERR_INVALID_ARG; }
    MI_TRY(icp_check_params(params));
    if (params->sum_mode == MI_SUM_CPU_SEQUENTIAL && c->distributed()) { set_error("mi_icp_load: MI_SUM_CPU_SEQUENTIAL needs a single-GPU context (the running sums follow one global point order)"); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    // mi_icp_load_times: host wall time per stage; with profiling on, the stream is drained at every mark
    const double t_begin = wall_ms();
    double t_mark = t_begin, a_mark = alloc_ms_counter();
    for (double& v : c->load_ms) v = 0.0;
    auto mark = [&](int stage) -> int {             // time since the last mark -> `stage`, its device allocations -> stage 0
        if (c->profile) MI_HIP(hipStreamSynchronize(c->stream));
        const double now = wall_ms(), a_now = alloc_ms_counter();
        c->load_ms[0] += a_now - a_mark;
        c->load_ms[stage] += (now - t_mark) - (a_now - a_mark);
        t_mark = now; a_mark = a_now;
        return MI_OK;
    };
    c->icp_loaded = false;
    c->icp = *params;
    // what the ranks split (mi_slam.h MI_SHARD_*): decided from GLOBAL sizes, so every rank decides alike
    c->source_sharded = false;
    if (c->distributed()) {   // (a one-rank communicator takes the same path: that is what the single-GPU box can test)
        const int per_rank = n_after / c->world;
        const bool indexed = resolve_nn_mode(c, params->nn_mode, params->shard_mode == MI_SHARD_TARGET ? per_rank : n_after) != MI_NN_BRUTEFORCE;
        c->source_sharded = params->shard_mode == MI_SHARD_SOURCE || (params->shard_mode == MI_SHARD_AUTO && indexed);
    }
    // Source sharding.  A rank's moving points should be (a) spatially DENSE wave by wave -- a wave's 64 points walk the box
    // hierarchy together, and sparse waves walk longer -- and (b) the same mix of easy and hard regions on every rank.  So every
    // rank orders the WHOLE moving cloud along the Hilbert curve (1 ms) and keeps the 64-point chunks rank, rank + W, rank + 2W ...
    // of that order.  (Clouds too small for a few chunks per rank are cut into contiguous slices of the caller's order instead.)
    const int n_all = n_before;
    c->n_global = n_all;
    const bool deal_chunks = c->source_sharded && n_all >= 4 * ICP_CHUNK_POINTS * c->world;
    if (c->source_sharded && !deal_chunks) {
        int slo = 0, shi = 0;
        shard_range(n_before, c->rank, c->world, &slo, &shi);
        before_xyz += 3 * (size_t)slo;
        n_before = shi - slo;
    }
    if (deal_chunks) MI_TRY(mi_source_share(n_all, c->rank, c->world, &n_before));
    c->n = n_before;
    c->n_pad = round_up(n_before, NN_SRC_PAD);
    const size_t np = (size_t)c->n_pad;
    MI_TRY(c->bx.reserve(np)); MI_TRY(c->by.reserve(np)); MI_TRY(c->bz.reserve(np));
    MI_TRY(c->keys.reserve(np));
    MI_TRY(reserve_rows(c));
    MI_TRY(mark(0));
    if (deal_chunks) {
        const int all_pad = round_up(n_all, NN_SRC_PAD);
        MI_TRY(c->cx.reserve((size_t)all_pad)); MI_TRY(c->cy.reserve((size_t)all_pad)); MI_TRY(c->cz.reserve((size_t)all_pad));
        MI_TRY(c->ax.reserve((size_t)all_pad)); MI_TRY(c->ay.reserve((size_t)all_pad)); MI_TRY(c->az.reserve((size_t)all_pad));
        MI_TRY(upload_soa(c, before_xyz, n_all, all_pad, c->cx.p, c->cy.p, c->cz.p, nullptr));
        MI_TRY(mark(1));
        MI_TRY(sort_sources(c, c->cx.p, c->cy.p, c->cz.p, n_all, all_pad, c->ax.p, c->ay.p, c->az.p));
        MI_HIP(deal_chunks_soa(c->ax.p, c->ay.p, c->az.p, n_all, c->rank, c->world, c->n, c->n_pad, c->bx.p, c->by.p, c->bz.p, c->stream));
        MI_TRY(mark(2));
    } else {
        MI_TRY(c->cx.reserve(np)); MI_TRY(c->cy.reserve(np)); MI_TRY(c->cz.reserve(np));
        // moving cloud: upload in the caller's order (cx.. as scratch), keep it Hilbert-sorted in bx..
        MI_TRY(upload_soa(c, before_xyz, n_before, c->n_pad, c->cx.p, c->cy.p, c->cz.p, nullptr));
        MI_TRY(mark(1));
        MI_TRY(sort_sources(c, c->cx.p, c->cy.p, c->cz.p, n_before, c->n_pad, c->bx.p, c->by.p, c->bz.p));
        MI_TRY(mark(2));
    }
    if (params->sum_mode == MI_SUM_CPU_SEQUENTIAL) {   // the sequential sums run in the CALLER's point order
        MI_TRY(c->sinv.reserve((size_t)n_before));
        MI_TRY(c->resid.reserve(np));
        MI_HIP(invert_order(c->sorder.p, n_before, c->sinv.p, c->stream));
    }
    MI_TRY(mark(2));
    // The FIXED cloud's share of the load -- upload, box hierarchy, cell grid -- depends on nothing the moving cloud's does, so it runs
    // on its own lanes (round 4): the upload and the hierarchy on `aux`, the grid (behind the upload) on `aux2`, their own scratch sets,
    // while `stream` is still ordering the moving cloud.  The host only copies into the pinned ring and enqueues; its one wait -- the grid's
    // bounding box -- is a wait for the fixed cloud's upload, which it would have sat out anyway.  `stream` then waits for both lanes.
    // (With profiling on, everything stays on `stream`, stage by stage: mi_icp_load_times drains it at every mark.)
    const bool lanes = !c->profile && c->aux != nullptr && c->aux2 != nullptr;
    const int m_local_pre = [&] { int lo = 0, hi = n_after; if (!c->source_sharded) shard_range(n_after, c->rank, c->world, &lo, &hi); return hi - lo; }();
    // The grid search carries the whole O(N) part of the iteration (nn_grid.hip) unless a stand-alone step has to come between
    // the search and the sums: the key all-reduce of a sharded fixed cloud, or cpu-slam's sequential running sums.
    const int mode = resolve_nn_mode(c, params->nn_mode, m_local_pre);
    // Every way out of this function joins the lanes into `stream`: an early return between here and the join below -- the
    // hierarchy refusing an over-tall cloud, a failed reserve -- used to leave work on aux / aux2 that `stream`, the only stream the next call's
    // buffer retirement and ~CtxScope look at, knew nothing about.
    struct LaneJoin {
        mi_ctx* c; bool on; bool joined = false;
        ~LaneJoin()
        {
            if (!on || joined) return;
            if (hipEventRecord(c->aux_event[1], c->aux) == hipSuccess) (void)hipStreamWaitEvent(c->stream, c->aux_event[1], 0);
            if (hipEventRecord(c->aux_event[2], c->aux2) == hipSuccess) (void)hipStreamWaitEvent(c->stream, c->aux_event[2], 0);
        }
    } lane_join{c, lanes};
    {
        LaneScope ls(c, lanes ? c->aux : nullptr, lanes ? 1 : 0);
        MI_TRY(upload_target_shard(c, after_xyz, n_after, c->source_sharded));
        if (lanes) MI_HIP(hipEventRecord(c->aux_event[0], c->aux));           // the fixed cloud is on the device
    }
    MI_TRY(mark(3));
    const int m_local = c->shard_hi - c->shard_lo;
    c->fused = mode == MI_NN_GRID && (!c->distributed() || c->source_sharded) && params->sum_mode == MI_SUM_EXACT;
    if (mode != MI_NN_BRUTEFORCE) {          // build the indexes now, not inside the first timed iteration
        {
            LaneScope ls(c, lanes ? c->aux : nullptr, lanes ? 1 : 0);
            MI_TRY(ensure_tree(c, m_local, c->shard_lo));
        }
        MI_TRY(mark(4));
        if (mode == MI_NN_GRID) {
            if (lanes) MI_HIP(hipStreamWaitEvent(c->aux2, c->aux_event[0], 0));
            LaneScope ls(c, lanes ? c->aux2 : nullptr, 0);                    // (the grid build has its own scratch; lane 0's sort scratch is not touched)
            MI_TRY(ensure_grid(c, m_local, c->shard_lo));
        }
        MI_TRY(mark(5));
    }
    if (lanes) {                             // everything after the load runs on `stream`: it waits for both lanes here, once
        MI_HIP(hipEventRecord(c->aux_event[1], c->aux));
        MI_HIP(hipEventRecord(c->aux_event[2], c->aux2));
        MI_HIP(hipStreamWaitEvent(c->stream, c->aux_event[1], 0));
        MI_HIP(hipStreamWaitEvent(c->stream, c->aux_event[2], 0));
        lane_join.joined = true;
    }
    c->icp_loaded = true;
    MI_TRY(mi_icp_reset(c));
    MI_TRY(mark(6));
    c->load_ms[7] = wall_ms() - t_begin;
    return MI_OK;
}

extern "C" int mi_icp_load_times(mi_ctx* c, double out_ms[MI_LOAD_STAGES])
{
    if (!c || !out_ms) { set_error("mi_icp_load_times: null argument"); return MI_ERR_INVALID_ARG; }
    for (int i = 0; i < MI_LOAD_STAGES; i++) out_ms[i] = c->load_ms[i];
    return MI_OK;
}

static IcpRules icp_rules(const mi_ctx* c)
{
    IcpRules rules{};
    rules.eps = c->icp.eps;
    rules.max_iterations = c->icp.max_iterations;
    rules.filter_pairs = c->icp.filter_pairs;
    rules.abort_on_increase = c->icp.abort_on_increase;
    rules.m_total = c->m_total;
    rules.seq_sums = c->icp.sum_mode == MI_SUM_CPU_SEQUENTIAL;
    rules.svd_ieee = c->tune.svd_ieee;
    return rules;
}

// The error sums of the last enqueued iteration have not been turned into its stop rule yet (they normally ride with the next
// iteration's moments).  Called when the host stops enqueuing and wants the state.
static int icp_flush_pending(mi_ctx* c)
{
    const IcpView v = make_view(c);
    const int nrows = icp_row_count(c->n);
    const int reduced = icp_reduced_count(nrows);
    // fused path: nobody has evaluated the last applied transform yet -- the next search would have
    if (c->fused) { ProfScope ps(c, MI_KERNEL_TRANSFORM); MI_HIP(icp_transform_error_rows(v, c->rows.p, 0, c->stream)); }
    ProfScope ps(c, MI_KERNEL_FINALIZE);
    MI_HIP(icp_rows_reduce(c->rows.p, nrows, c->rows_reduced.p, c->stream));
    if (c->distributed()) {
        MI_HIP(icp_rows_to_state(c->d_state, c->rows_reduced.p, reduced, 2, c->stream));
        MI_TRY(allreduce_doubles(c, c->d_state->err, ICP_ERRSUMS));
        MI_HIP(icp_finalize_pending(c->d_state, nullptr, 0, icp_rules(c), c->stream));
    } else {
        MI_HIP(icp_finalize_pending(c->d_state, c->rows_reduced.p, reduced, icp_rules(c), c->stream));
    }
    return MI_OK;
}

// One loop body of basicicp.cpp:32-57 / icpcuda.cu:31-54, enqueued without host synchronisation.  Three launches on the
// default path: fused search (transform, previous error, search, moments) -> rows reduce -> solve (previous stop rule, Kabsch,
// compose).
static int icp_enqueue_iteration(mi_ctx* c)
{
    const IcpView v = make_view(c);
    const int m_local = c->shard_hi - c->shard_lo;
    const IcpRules rules = icp_rules(c);
    const int nrows = icp_row_count(c->n);
    const int reduced = icp_reduced_count(nrows);
    const int seq = c->icp.sum_mode == MI_SUM_CPU_SEQUENTIAL;
    if (c->fused) {
        GridSearchArgs a{};
        a.n = c->n; a.keys = c->keys.p;
        a.stats = c->nn_stats_on ? c->nn_stats.p : nullptr;
        a.state = c->d_state;
        a.bx = c->bx.p; a.by = c->by.p; a.bz = c->bz.p;
        a.match_slot = c->match_slot.p; a.shard_lo = c->shard_lo; a.shard_hi = c->shard_hi;
        a.filter_pairs = c->icp.filter_pairs; a.max_distance_squared = c->icp.max_distance_squared;
        a.rows = c->rows.p;
        a.order = c->sched_order.p; a.far = c->sched_far.p; a.far_lanes = c->sched_lanes.p;
        a.deal_rows = c->tune.grid_deal_rows < 0 ? (c->n >= GRID_DEAL_ROWS_MIN_POINTS ? 1 : 0) : c->tune.grid_deal_rows;
        // (the host's own count of the iterations it has enqueued since the load / reset: the device's `passes` as long as the registration runs)
        a.extend_reach = c->enqueued_passes < GRID_COLD_PASSES ? 1 : 0;
        a.extend_reach_next = c->enqueued_passes + 1 < GRID_COLD_PASSES ? 1 : 0;
        a.split_walks = c->tune.grid_split_walks < 0 ? (c->n <= GRID_HELPER_FULL_MAX_POINTS ? 1 : (c->n <= GRID_HELPER_MAX_POINTS ? 2 : 0)) : c->tune.grid_split_walks;
        hipEvent_t e0 = nullptr, e1 = nullptr;           // timed, if at all, by events attached to the launch itself (nn_grid_query)
        MI_TRY(c->prof_span(MI_KERNEL_NN, &e0, &e1));
        MI_HIP(nn_grid_query(c->grid, c->tree, a, v.fma, c->stream, e0, e1));
        c->enqueued_passes += 1;
    } else {
        // K1 (+ C1), K2
        MI_TRY(launch_nn(c, c->cx.p, c->cy.p, c->cz.p, c->n, m_local, c->shard_lo, v.fma, &c->d_state->done, c->icp.nn_mode));
        if (!c->source_sharded) MI_TRY(allreduce_keys(c, c->n));   // source-sharded ranks hold disjoint moving points: nothing to merge
        { ProfScope ps(c, MI_KERNEL_MOMENTS); MI_HIP(icp_moments_rows(v, c->rows.p, c->stream)); }
        if (seq) { ProfScope ps(c, MI_KERNEL_MOMENTS); MI_HIP(icp_seq_centroids(v, c->stream)); }
    }
    {   // K3 (+ K6 of the previous iteration)
        const IcpSchedule sched = make_schedule(c);
        int* cursors = c->fused ? sched.counters : nullptr;
        if (c->distributed()) {
            // ONE all-reduce per iteration: this iteration's 16 moments and the previous iteration's 2 error sums ride together -- as
            // the REDUCED ROWS themselves (64 x 18 doubles, the rows past this rank's own count written as zeros EVERY iteration -- the
            // collective is in place, and a rank with fewer rows than its neighbour would otherwise re-send the neighbour's sums: every
            // rank sends the same length whatever its share), so that the solve kernel adds them up exactly as on one GPU and no kernel sits between
            // the reduction and the collective (a 9 KB all-reduce is as latency-bound as a 144-byte one)
            (void)reduced;
            { ProfScope ps(c, MI_KERNEL_SOLVE); MI_HIP(icp_rows_reduce(c->rows.p, nrows, c->rows_reduced.p, c->stream, c->fused ? &sched : nullptr, true)); }
            { ProfScope ps(c, MI_KERNEL_ALLREDUCE); MI_TRY(allreduce_doubles(c, c->rows_reduced.p, ICP_REDUCED_ROWS * (ICP_MOMENTS + ICP_ERRSUMS))); }
            ProfScope ps(c, MI_KERNEL_SOLVE);
            MI_HIP(icp_solve_deferred(c->d_state, c->rows_reduced.p, ICP_REDUCED_ROWS, c->icp.compose_mode, rules, 1, c->stream, cursors));
        } else if (nrows <= ICP_FUSED_SOLVE_MAX_ROWS && c->tune.icp_fused_solve != 0) {
            // small clouds: one launch of one workgroup for both (and no work order: every wave of such a search is resident from the start)
            ProfScope ps(c, MI_KERNEL_SOLVE);
            MI_HIP(icp_reduce_solve(c->d_state, c->rows.p, nrows, c->icp.compose_mode, rules, 1, c->stream));
        } else {
            ProfScope ps(c, MI_KERNEL_SOLVE);
            MI_HIP(icp_rows_reduce(c->rows.p, nrows, c->rows_reduced.p, c->stream, c->fused ? &sched : nullptr));
            MI_HIP(icp_solve_deferred(c->d_state, c->rows_reduced.p, reduced, c->icp.compose_mode, rules, 1, c->stream, cursors));
        }
    }
    if (!c->fused) {
        // K4+K5: its error sums wait in the rows for the next solve (or the flush)
        { ProfScope ps(c, MI_KERNEL_TRANSFORM); MI_HIP(icp_transform_error_rows(v, c->rows.p, 2, c->stream)); }
        if (seq) { ProfScope ps(c, MI_KERNEL_TRANSFORM); MI_HIP(icp_seq_error(v, c->stream)); }
    }
    return MI_OK;
}

static int icp_fetch_state(mi_ctx* c)
{
    { StallProbe sp("fetch_state: enqueue copy"); MI_HIP(hipMemcpyAsync(c->h_state, c->d_state, sizeof(IcpState), hipMemcpyDeviceToHost, c->stream)); }
    { StallProbe sp("fetch_state: stream synchronize"); MI_HIP(hipStreamSynchronize(c->stream)); }
    { StallProbe sp("fetch_state: retire buffers"); retire_buffers(c); }
    return MI_OK;
}

extern "C" int mi_icp_auto_batch(long long n_moving_total, long long m_fixed_total, int world, int source_sharded, int every_pair_search)
{
    if (world < 1) world = 1;
    const double n_rank = (double)n_moving_total / world, m_rank = (double)m_fixed_total / world;
    const double est_s = every_pair_search ? (source_sharded ? n_rank * (double)m_fixed_total : (double)n_moving_total * m_rank) / 7e12
                                           : 2e-5 + 5e-11 * (source_sharded ? n_rank : (double)n_moving_total);
    return est_s >= 5e-3 ? 1 : (est_s >= 2e-4 ? 4 : (est_s >= 1e-4 ? 8 : 16));
}

extern "C" int mi_icp_run(mi_ctx* c, int max_new_iterations, int* iterations_done)
{
    if (!c || !c->icp_loaded) { set_error("mi_icp_run: no problem loaded"); return MI_ERR_STATE; }
    MI_ENTER(c);
    MI_TRY(icp_fetch_state(c));
    const int passes_before = c->h_state->passes;
    int batch = c->icp.sync_every;
    if (batch <= 0) {
        // auto: a long iteration dwarfs a host round trip (check after each one); short ones are launch-bound (batch them).
        // Estimated from the measured rates: every-pair ~7e12 pairs/s, indexed searches ~5e-11 s per moving point + launches.
        // From GLOBAL sizes only: every batch ends in a collective, so all ranks must pick the same batch (a rank's own share
        // of the dealt moving cloud differs from its neighbours' by up to 64 points).
        const bool brute = resolve_nn_mode(c, c->icp.nn_mode, c->source_sharded ? c->m_total : c->m_total / c->world) == MI_NN_BRUTEFORCE;
        batch = mi_icp_auto_batch(c->n_global, c->m_total, c->world, c->source_sharded ? 1 : 0, brute ? 1 : 0);
    }
    if (c->icp.verbose) batch = 1;       // one "loop_nr" line per iteration, like basicicp.cpp:50 / icpcuda.cu:39
    int enqueued = 0;
    // Batches of more than one iteration are PIPELINED (round 4): an intermediate host check used to settle the pending iteration (two
    // launches over the moving cloud, ~23 us at 1e6 points), copy the state and drain the stream -- ~50 us in which the device waits for the
    // host to read one flag and enqueue the next batch.  Now a check only PEEKS: the state block is copied behind the batch, ONE iteration
    // of the next batch goes onto the stream behind the copy, and the host waits for the copy alone; the device is never idle, and the
    // deferred stop rule makes the peek sound (a stop shows up one iteration late and that iteration applies nothing -- exactly as inside
    // a batch; an iteration enqueued behind a stop returns at once).  The pending iteration is settled once, when the loop ends.
    const bool pipelined = batch > 1 && !c->profile && g_stall_ms <= 0 && c->peek_event != nullptr && c->tune.icp_pipeline != 0;
    bool ahead = false;                  // one iteration of the next batch is on the stream already
    while (pipelined && !c->h_state->done && (max_new_iterations < 0 || enqueued < max_new_iterations)) {
        int todo = batch;
        if (max_new_iterations >= 0) todo = std::min(todo, max_new_iterations - enqueued + (ahead ? 1 : 0));
        if (c->icp.max_iterations >= 0) todo = std::max(1, std::min(todo, c->icp.max_iterations - c->h_state->iterations));
        for (int b = ahead ? 1 : 0; b < todo; b++) { MI_TRY(icp_enqueue_iteration(c)); enqueued++; }
        const bool last = (max_new_iterations >= 0 && enqueued >= max_new_iterations) ||
                          (c->icp.max_iterations >= 0 && c->h_state->iterations + todo >= c->icp.max_iterations);
        if (last) { ahead = false; break; }                           // (the budget is spent: nothing to decide, settle below)
        MI_HIP(hipMemcpyAsync(c->h_state, c->d_state, sizeof(IcpState), hipMemcpyDeviceToHost, c->stream));
        MI_HIP(hipEventRecord(c->peek_event, c->stream));
        MI_TRY(icp_enqueue_iteration(c));                             // the next batch's first iteration, behind the copy
        enqueued++;
        ahead = true;
        MI_HIP(hipEventSynchronize(c->peek_event));
    }
    if (pipelined) {
        MI_TRY(icp_flush_pending(c));
        MI_TRY(icp_fetch_state(c));
    }
    while (!pipelined && !c->h_state->done && (max_new_iterations < 0 || enqueued < max_new_iterations)) {
        int todo = batch;
        if (max_new_iterations >= 0) todo = std::min(todo, max_new_iterations - enqueued);
        // a run capped at max_iterations is not enqueued past the cap (the device would turn the surplus into launches that return at once:
        // 14 of them behind a 50-iteration run in batches of 16)
        if (c->icp.max_iterations >= 0) todo = std::max(1, std::min(todo, c->icp.max_iterations - c->h_state->iterations));
        hipEvent_t dev_e0 = nullptr, dev_e1 = nullptr;
        if (g_stall_ms > 0) { (void)hipEventCreate(&dev_e0); (void)hipEventCreate(&dev_e1); (void)hipEventRecord(dev_e0, c->stream); }
        { StallProbe sp("run: enqueue batch"); for (int b = 0; b < todo; b++) MI_TRY(icp_enqueue_iteration(c)); }
        { StallProbe sp("run: enqueue flush"); MI_TRY(icp_flush_pending(c)); }
        if (dev_e1) (void)hipEventRecord(dev_e1, c->stream);
        enqueued += todo;
        const int shown = c->h_state->passes;
        MI_TRY(icp_fetch_state(c));
        if (dev_e1) {
            float dev_ms = 0.f;
            (void)hipEventElapsedTime(&dev_ms, dev_e0, dev_e1);
            if (dev_ms > g_stall_ms) fprintf(stderr, "mislam stall: the batch's %d iterations took %.2f ms ON THE DEVICE (events)\n", todo, dev_ms);
            (void)hipEventDestroy(dev_e0); (void)hipEventDestroy(dev_e1);
        }
        if (c->icp.verbose && c->rank == 0 && c->h_state->passes > shown)
            printf("loop_nr %d, error: %f, correspondencesSize: %d\n", c->h_state->passes - 1, c->h_state->error, c->h_state->pairs);
    }
    if (iterations_done) *iterations_done = c->h_state->passes - passes_before;
    return MI_OK;
}

extern "C" int mi_icp_result(mi_ctx* c, float out_T[16], int* iterations, float* error, int* stop_reason)
{
    if (!c || !c->icp_loaded) { set_error("mi_icp_result: no problem loaded"); return MI_ERR_STATE; }
    MI_ENTER(c);
    MI_TRY(icp_fetch_state(c));
    const IcpState* s = c->h_state;
    if (out_T) {
        // column-major 4x4, t in column 3 (ConvertToTransformationMatrix, common.cpp:353-358)
        for (int col = 0; col < 3; col++) {
            for (int row = 0; row < 3; row++) out_T[4 * col + row] = s->R[3 * col + row];
            out_T[4 * col + 3] = 0.f;
        }
        out_T[12] = s->t[0]; out_T[13] = s->t[1]; out_T[14] = s->t[2]; out_T[15] = 1.f;
    }
    if (iterations) *iterations = s->iterations;
    if (error) *error = s->error;
    if (stop_reason) *stop_reason = s->done ? s->stop_reason : MI_STOP_RUNNING;
    return MI_OK;
}

extern "C" int mi_icp_register(mi_ctx* c, const float* before_xyz, int n_before, const float* after_xyz, int n_after,
                               const mi_icp_params* params, float out_T[16], int* iterations, float* error)
{
    if (!iterations || !error || !out_T) { set_error("mi_icp_register: out_T, iterations and error must be non-null"); return MI_ERR_INVALID_ARG; }
    MI_TRY(mi_icp_load(c, before_xyz, n_before, after_xyz, n_after, params));
    MI_TRY(mi_icp_run(c, -1, nullptr));
    return mi_icp_result(c, out_T, iterations, error, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------
// test-grade primitives
// ---------------------------------------------------------------------------------------------------------------
extern "C" int mi_nn_search(mi_ctx* c, const float* src_xyz, int n, const float* tgt_xyz, int m, int dist_mode, int* idx, float* d2)
{
    return mi_nn_search_ex(c, src_xyz, n, tgt_xyz, m, dist_mode, MI_NN_AUTO, idx, d2);
}

extern "C" int mi_nn_search_ex(mi_ctx* c, const float* src_xyz, int n, const float* tgt_xyz, int m, int dist_mode, int nn_mode,
                               int* idx, float* d2)
{
    if (!c) { set_error("mi_nn_search: null context"); return MI_ERR_INVALID_ARG; }
    if (nn_mode != MI_NN_AUTO && nn_mode != MI_NN_BRUTEFORCE && nn_mode != MI_NN_TREE && nn_mode != MI_NN_GRID) { set_error("mi_nn_search: bad nn_mode"); return MI_ERR_INVALID_ARG; }
    if (n < 0 || m < 0 || (n > 0 && (!src_xyz || !idx)) || (m > 0 && !tgt_xyz)) { set_error("mi_nn_search: bad arguments"); return MI_ERR_INVALID_ARG; }
    if (dist_mode != MI_DIST_CPU_ROUNDING && dist_mode != MI_DIST_FMA) { set_error("mi_nn_search: bad dist_mode"); return MI_ERR_INVALID_ARG; }
    if (n == 0) return MI_OK;
    if (m == 0) { set_error("mi_nn_search: empty target cloud"); return MI_ERR_INVALID_ARG; }
    if (m < c->world) { set_error("mi_nn_search: fewer targets than ranks"); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    c->icp_loaded = false;   // the workspace is being reused
    const int n_pad = round_up(n, NN_SRC_PAD);
    MI_TRY(c->bx.reserve((size_t)n_pad)); MI_TRY(c->by.reserve((size_t)n_pad)); MI_TRY(c->bz.reserve((size_t)n_pad));
    MI_TRY(c->cx.reserve((size_t)n_pad)); MI_TRY(c->cy.reserve((size_t)n_pad)); MI_TRY(c->cz.reserve((size_t)n_pad));
    MI_TRY(c->keys.reserve((size_t)n_pad));
    MI_TRY(upload_soa(c, src_xyz, n, n_pad, c->bx.p, c->by.p, c->bz.p, nullptr));
    MI_TRY(sort_sources(c, c->bx.p, c->by.p, c->bz.p, n, n_pad, c->cx.p, c->cy.p, c->cz.p));
    MI_TRY(upload_target_shard(c, tgt_xyz, m));
    MI_HIP(fill_keys(c->keys.p, n, c->stream));
    MI_TRY(launch_nn(c, c->cx.p, c->cy.p, c->cz.p, n, c->shard_hi - c->shard_lo, c->shard_lo, dist_mode == MI_DIST_FMA, nullptr, nn_mode));
    MI_TRY(allreduce_keys(c, n));
    MI_TRY(c->idx_tmp.reserve((size_t)n));
    MI_TRY(c->staging.reserve((size_t)n));
    // keys are in sorted-slot order: scatter back to the caller's order
    MI_HIP(unpack_keys(c->keys.p, c->sorder.p, n, c->idx_tmp.p, d2 ? c->staging.p : nullptr, c->stream));
    MI_HIP(hipMemcpyAsync(idx, c->idx_tmp.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (d2) MI_HIP(hipMemcpyAsync(d2, c->staging.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipStreamSynchronize(c->stream));
    return MI_OK;
}

// Shared set-up of mi_kabsch / mi_transform_mse: source -> (b, c), target -> tgt4, caller's correspondences -> keys.
static int load_pairs(mi_ctx* c, const float* src_xyz, int n, const float* tgt_xyz, int m, const int* idx, const unsigned char* keep)
{
    MI_ENTER(c);
    c->icp_loaded = false;
    if (c->world != 1) { set_error("this primitive is single-GPU only"); return MI_ERR_STATE; }
    for (int i = 0; idx && i < n; i++)
        if (idx[i] < 0 || idx[i] >= m) { set_error("correspondence idx[%d] = %d outside [0,%d)", i, idx[i], m); return MI_ERR_INVALID_ARG; }
    c->n = n;
    c->n_pad = round_up(n, NN_SRC_PAD);
    const size_t np = (size_t)c->n_pad;
    MI_TRY(c->bx.reserve(np)); MI_TRY(c->by.reserve(np)); MI_TRY(c->bz.reserve(np));
    MI_TRY(c->cx.reserve(np)); MI_TRY(c->cy.reserve(np)); MI_TRY(c->cz.reserve(np));
    MI_TRY(c->keys.reserve(np));
    MI_TRY(reserve_rows(c));
    MI_TRY(upload_soa(c, src_xyz, n, c->n_pad, c->bx.p, c->by.p, c->bz.p, nullptr));
    const size_t bytes = sizeof(float) * np;
    MI_HIP(hipMemcpyAsync(c->cx.p, c->bx.p, bytes, hipMemcpyDeviceToDevice, c->stream));
    MI_HIP(hipMemcpyAsync(c->cy.p, c->by.p, bytes, hipMemcpyDeviceToDevice, c->stream));
    MI_HIP(hipMemcpyAsync(c->cz.p, c->bz.p, bytes, hipMemcpyDeviceToDevice, c->stream));
    if (tgt_xyz && m > 0) MI_TRY(upload_target_shard(c, tgt_xyz, m));
    else { c->m_total = 0; c->shard_lo = c->shard_hi = 0; MI_TRY(c->tgt4.reserve(1)); }
    if (idx) {
        MI_TRY(c->idx_tmp.reserve((size_t)n));
        MI_HIP(hipMemcpyAsync(c->idx_tmp.p, idx, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, c->stream));
        if (keep) {
            MI_TRY(c->keep_tmp.reserve((size_t)n));
            MI_HIP(hipMemcpyAsync(c->keep_tmp.p, keep, (size_t)n, hipMemcpyHostToDevice, c->stream));
        }
        MI_HIP(pack_keys(c->idx_tmp.p, keep ? c->keep_tmp.p : nullptr, n, c->keys.p, c->stream));
    } else {
        MI_HIP(fill_keys(c->keys.p, n, c->stream));   // index 0xFFFFFFFF is outside every shard: no pair is used
    }
    mi_icp_params_default(&c->icp);
    c->icp.filter_pairs = 1;                       // keys carry d2 = 0 (kept) or +inf (dropped)
    c->icp.max_distance_squared = 1.f;
    state_identity(c->h_state);
    return MI_OK;
}

extern "C" int mi_kabsch(mi_ctx* c, const float* src_xyz, int n, const float* tgt_xyz, int m, const int* idx,
                         const unsigned char* keep, float out_R9[9], float out_t3[3], int* pairs_used)
{
    if (!c || !src_xyz || !tgt_xyz || !idx || n <= 0 || m <= 0 || !out_R9 || !out_t3) { set_error("mi_kabsch: bad arguments"); return MI_ERR_INVALID_ARG; }
    MI_TRY(load_pairs(c, src_xyz, n, tgt_xyz, m, idx, keep));
    MI_HIP(hipMemcpyAsync(c->d_state, c->h_state, sizeof(IcpState), hipMemcpyHostToDevice, c->stream));
    const IcpView v = make_view(c);
    const int nrows = icp_row_count(n);
    { ProfScope ps(c, MI_KERNEL_MOMENTS); MI_HIP(icp_moments_rows(v, c->rows.p, c->stream)); }
    {
        ProfScope ps(c, MI_KERNEL_SOLVE);
        MI_HIP(icp_rows_reduce(c->rows.p, nrows, c->rows_reduced.p, c->stream));
        MI_HIP(icp_solve_deferred(c->d_state, c->rows_reduced.p, icp_reduced_count(nrows), MI_COMPOSE_EXACT, IcpRules{}, 0, c->stream));
    }
    MI_TRY(icp_fetch_state(c));
    if (pairs_used) *pairs_used = c->h_state->pairs;
    if (c->h_state->pairs <= 0) { set_error("mi_kabsch: no pair kept"); return MI_ERR_INVALID_ARG; }
    memcpy(out_R9, c->h_state->Ri, sizeof(float) * 9);
    memcpy(out_t3, c->h_state->ti, sizeof(float) * 3);
    return MI_OK;
}

extern "C" int mi_cross_moments(mi_ctx* c, const float* src_xyz, int n, const float* tgt_xyz, int m, const int* idx,
                                const unsigned char* keep, double out16[16])
{
    if (!c || !src_xyz || !tgt_xyz || !idx || n <= 0 || m <= 0 || !out16) { set_error("mi_cross_moments: bad arguments"); return MI_ERR_INVALID_ARG; }
    MI_TRY(load_pairs(c, src_xyz, n, tgt_xyz, m, idx, keep));
    MI_HIP(hipMemcpyAsync(c->d_state, c->h_state, sizeof(IcpState), hipMemcpyHostToDevice, c->stream));
    const IcpView v = make_view(c);
    const int nrows = icp_row_count(n);
    { ProfScope ps(c, MI_KERNEL_MOMENTS); MI_HIP(icp_moments_rows(v, c->rows.p, c->stream)); }
    MI_HIP(icp_rows_reduce(c->rows.p, nrows, c->rows_reduced.p, c->stream));
    MI_HIP(icp_rows_to_state(c->d_state, c->rows_reduced.p, icp_reduced_count(nrows), 1, c->stream));
    MI_TRY(icp_fetch_state(c));
    for (int i = 0; i < ICP_MOMENTS; i++) out16[i] = c->h_state->mom[i];
    return MI_OK;
}

extern "C" int mi_transform_mse(mi_ctx* c, const float* src_xyz, int n, const float R9[9], const float t3[3],
                                const float* tgt_xyz, int m, const int* idx, const unsigned char* keep, int divide_by_pairs,
                                float* out_xyz, float* mse)
{
    if (!c || !src_xyz || n <= 0 || !R9 || !t3) { set_error("mi_transform_mse: bad arguments"); return MI_ERR_INVALID_ARG; }
    if (mse && (!tgt_xyz || !idx || m <= 0)) { set_error("mi_transform_mse: mse needs target cloud and idx"); return MI_ERR_INVALID_ARG; }
    MI_TRY(load_pairs(c, src_xyz, n, tgt_xyz, m, mse ? idx : nullptr, keep));
    memcpy(c->h_state->R, R9, sizeof(float) * 9);
    memcpy(c->h_state->t, t3, sizeof(float) * 3);
    MI_HIP(hipMemcpyAsync(c->d_state, c->h_state, sizeof(IcpState), hipMemcpyHostToDevice, c->stream));
    const IcpView v = make_view(c);
    const int nrows = icp_row_count(n);
    { ProfScope ps(c, MI_KERNEL_TRANSFORM); MI_HIP(icp_transform_error_rows(v, c->rows.p, 0, c->stream)); }
    MI_HIP(icp_rows_reduce(c->rows.p, nrows, c->rows_reduced.p, c->stream));
    MI_HIP(icp_rows_to_state(c->d_state, c->rows_reduced.p, icp_reduced_count(nrows), 2, c->stream));
    if (out_xyz) {
        MI_TRY(c->staging.reserve((size_t)3 * n));
        MI_HIP(soa_to_aos(c->cx.p, c->cy.p, c->cz.p, n, c->staging.p, c->stream));
        MI_HIP(hipMemcpyAsync(out_xyz, c->staging.p, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    }
    MI_TRY(icp_fetch_state(c));
    if (mse) {
        const double denom = divide_by_pairs ? c->h_state->err[1] : (double)m;
        *mse = (float)(c->h_state->err[0] / denom);
    }
    return MI_OK;
}
