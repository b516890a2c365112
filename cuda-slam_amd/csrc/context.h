// Context object behind the opaque mi_ctx handle: device, stream, grow-only workspace, optional RCCL communicator,
// HIP-event profiling.  Shared by the ICP and CPD drivers.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/mi_slam.h"
#include "kernels.h"
#include "nn_grid.h"
#include "nn_tree.h"

namespace mislam {

void set_error(const char* fmt, ...);

#define MI_HIP(call)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            mislam::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);  \
            return MI_ERR_HIP;                                                                             \
        }                                                                                                  \
    } while (0)

#define MI_NCCL(call)                                                                                      \
    do {                                                                                                   \
        ncclResult_t r_ = (call);                                                                          \
        if (r_ != ncclSuccess) {                                                                           \
            mislam::set_error("%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
            return MI_ERR_RCCL;                                                                            \
        }                                                                                                  \
    } while (0)

#define MI_TRY(call)             \
    do {                         \
        int rc_ = (call);        \
        if (rc_ != MI_OK) return rc_; \
    } while (0)

// Device buffers that were outgrown: work already enqueued may still read them, so they are released at the next point where the
// host has drained the stream anyway (retire_buffers, called behind the loads' and runs' own synchronisations) -- not behind a
// device-wide synchronisation per buffer, which is what a load of forty buffers used to pay when a size was new.
// The list is the CONTEXT's (mi_ctx::retired): a buffer outgrown inside a call on context A is released only behind a drain of A's own
// stream, with A's device current -- never by another host thread's context, never on another device (round 3 kept one process-wide list).
void retire_later(void* p);                        // into the list of the context whose call is running on this thread (CtxScope)
void retire_buffers(struct ::mi_ctx* ctx);         // call right behind a synchronisation of ctx->stream, ctx->device current
// Device memory comes out of the runtime's stream-ordered pool, kept whole (release threshold: never): hipFree of a plain
// allocation costs ~0.2 ms on this machine (tools/alloc_probe.cpp) -- forty buffers outgrown by a new size were 8 ms -- the pool's
// free is ~1 us and its memory is handed out again.  MISLAM_POOL=0 (or a runtime without the pool) falls back to hipMalloc / hipFree.
hipError_t device_alloc(void** p, size_t bytes);
void device_free(void* p);
double& alloc_ms_counter();        // host ms this thread has spent in hipMalloc through DevBuf::reserve (mi_icp_load_times)
double wall_ms();

// grow-only device buffer; grows by at least half (a sweep over slowly rising sizes reallocates a few times, not every call)
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    int reserve(size_t count)
    {
        if (count <= cap) return MI_OK;
        if (p) retire_later(p);
        p = nullptr;
        const size_t grown = cap + cap / 2;
        if (cap != 0 && count < grown) count = grown;
        cap = 0;
        const double t0 = wall_ms();
        MI_HIP(device_alloc((void**)&p, count * sizeof(T)));
        alloc_ms_counter() += wall_ms() - t0;
        cap = count;
        return MI_OK;
    }
    void release()
    {
        if (p) device_free(p);
        p = nullptr;
        cap = 0;
    }
};

struct ProfileSpan {
    int kernel;
    hipEvent_t e0, e1;
};

struct CpdWorkspace;   // cpd_api.hip

}  // namespace mislam

struct mi_ctx {
    int device = 0;
    std::vector<void*> retired;                          // outgrown device buffers waiting for the next drain of `stream` (retire_buffers)
    hipStream_t stream = nullptr;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    mi_exchange_fn exchange = nullptr;                   // the caller's transport instead of RCCL (mi_ctx_create_exchange)
    void* exchange_user = nullptr;
    void* exchange_host = nullptr;                       // pinned staging for it
    size_t exchange_cap = 0;
    bool distributed() const { return comm != nullptr || exchange != nullptr; }
    int cu_count = 256;
    // developer switches, read ONCE at context creation (none changes a result)
    struct Tuning {
        int nn_force_mode = 0;                           // MISLAM_NN_MODE: force MI_NN_* whatever the call asks
        int nn_R = 2, nn_wgs = 0, nn_chunks = 0;         // MISLAM_NN_R / _WGS / _CHUNKS: K1 sources per lane, workgroup budget, target chunks
        float grid_points_per_cell = mislam::GRID_POINTS_PER_CELL;   // MISLAM_GRID_PPC
        int cpd_mfma = 1;                                // MISLAM_CPD_MFMA=0: VALU contraction instead of MFMA
        int cpd_trunc_cull = 1;                          // MISLAM_CPD_TRUNC_CULL=0: the hybrid mode's truncated E-step over every pair (round 4's kernels) instead of K7t
        int fgt_resume = 1;                              // MISLAM_FGT_RESUME=0: re-cluster the fixed cloud from scratch every E-step
        int grid_split_walks = -1;                       // MISLAM_GRID_SPLIT_WALKS=0 / 1 / 2: the fused search's helper waves: none / for every walk / only beside a scan (default: by size)
        int grid_deal_rows = -1;                         // MISLAM_GRID_DEAL_ROWS=0 / 1: K1g's leftover rows never / always dealt out one per lane (default: by size)
        int svd_ieee = 0;                                // MISLAM_SVD_IEEE=1: the 3 x 3 SVD of every solve in IEEE divisions and roots (svd3.hpp)
        int icp_fused_solve = 1;                         // MISLAM_ICP_FUSED_SOLVE=0: rows reduce and solve as two launches at every size
        int icp_pipeline = 1;                            // MISLAM_ICP_PIPELINE=0: every host check of mi_icp_run settles the pending iteration and drains the stream
        int fgt_two_streams = 1;                         // MISLAM_FGT_TWO_STREAMS=0: the fixed cloud's clustering of an FGT E-step on the main stream, behind the moving side's
        int fgt_lists_in_model = 1;                      // MISLAM_FGT_LISTS_IN_MODEL=0: the member lists of an FGT E-step by the three-launch counting sort (round 4) instead of inside the model kernel
        int fgt_coop_sweep = 1;                          // MISLAM_FGT_COOP_SWEEP=0: K-centre sweeps of clouds beyond 16 384 points as in rounds 1-4 (one workgroup / two launches per centre); 2: the cooperative kernel for every sweep
        int fgt_shard_queries = 1;                       // MISLAM_FGT_SHARD_QUERIES=0: the FGT / hybrid CPD modes run replicated on a multi-rank context, no collective (rounds 4-5)
        int fgt_model_splits = 1;                        // MISLAM_FGT_MODEL_SPLITS=0: one workgroup per cell in the FGT model build whatever the cells' sizes (rounds 1-4)
        int fgt_replay = 1;                              // MISLAM_FGT_REPLAY=0: sweep the moving cloud step by step every E-step (no guess replayed)
    } tune;

    // ---- a second lane for the loads (round 4): mi_icp_load uploads and indexes the FIXED cloud on `aux` (hierarchy) and `aux2` (cell grid)
    // while the moving cloud is uploaded and ordered on `stream`; the helpers enqueue on work_stream() and take their scratch from the set
    // `lane` names (0: the buffers below, 1: the *2 copies), so two lanes never share a scratch buffer.  Everything else runs on `stream`.
    hipStream_t aux = nullptr, aux2 = nullptr;
    hipEvent_t aux_event[3] = {nullptr, nullptr, nullptr};
    hipStream_t lane_stream = nullptr;                   // null: `stream`
    int lane = 0;
    hipStream_t work_stream() const { return lane_stream ? lane_stream : stream; }
    mislam::DevBuf<float> staging2;
    mislam::DevBuf<unsigned int> tcodes2_in, tcodes2_out;
    mislam::DevBuf<int> torder2_in;
    mislam::DevBuf<float> tbbox2, gbbox;
    mislam::DevBuf<unsigned char> tsort_temp2;

    // ---- workspace shared by the drivers
    mislam::DevBuf<float> staging;                       // AoS upload/download staging
    // pinned host staging of the cloud uploads (host_to_device): the runtime's own pageable-copy path stalls for 20-50 ms every
    // few dozen calls on this machine (tools/alloc_probe.cpp: median 0.36 ms, max 27 ms for 12 MB).  A ring of PIN_SLOTS pieces of
    // 1 MB, pinned once per context: a piece is copied in while the pieces before it are on their way -- 0.30 ms for 12 MB, every time
    static constexpr int PIN_SLOTS = 16;
    static constexpr size_t PIN_PIECE = 1u << 20;
    char* pin = nullptr;
    hipEvent_t pin_event[PIN_SLOTS] = {nullptr};         // recorded behind the last transfer out of slot k
    unsigned int pin_busy = 0;                           // bit k: pin_event[k] has been recorded and not waited for since
    unsigned int pin_next = 0;
    float* h_scratch = nullptr;                          // 64 pinned floats for small read-backs
    mislam::DevBuf<float> bx, by, bz;                    // original moving cloud, SoA
    mislam::DevBuf<float> cx, cy, cz;                    // current (transformed) moving cloud, SoA
    mislam::DevBuf<float> ax, ay, az;                    // multi-GPU load scratch: the whole moving cloud in Hilbert order
    mislam::DevBuf<float> tx, ty, tz;                    // this rank's fixed-cloud shard, SoA (K1 scalar streams)
    mislam::DevBuf<float4> tgt4;                         // same shard as float4 (gathers)
    mislam::DevBuf<unsigned long long> keys;
    mislam::DevBuf<double> part_mom, part_err;           // per-workgroup partial sums of the NICP / CPD drivers
    mislam::DevBuf<int> sched_order, sched_counters;     // work order of the fused search (IcpSchedule)
    mislam::DevBuf<unsigned char> sched_far;
    mislam::DevBuf<unsigned long long> sched_lanes;      // per chunk: the lanes that walked (IcpSchedule::lanes)
    mislam::DevBuf<double> rows, rows_reduced;           // ICP: one row of 18 sums per 64 moving points (icp_rows.hpp), and <= 64 reduced rows
    mislam::DevBuf<int> idx_tmp;
    mislam::DevBuf<unsigned char> keep_tmp;
    mislam::IcpState* d_state = nullptr;
    mislam::IcpState* h_state = nullptr;                 // pinned
    hipEvent_t peek_event = nullptr;                     // behind the state copy of an intermediate host check (mi_icp_run)

    // ---- exact-NN indexes over the fixed-cloud shard (built lazily, valid until the shard is replaced)
    mislam::DevBuf<unsigned int> tcodes_in, tcodes_out;  // Morton sort scratch (also used for the moving cloud's ordering)
    mislam::DevBuf<int> torder_in, torder_out;
    mislam::DevBuf<float> tbbox;
    mislam::DevBuf<unsigned char> tsort_temp;
    mislam::DevBuf<float4> tpts, tboxes;                 // box hierarchy (nn_tree.h): sorted points and node boxes (build scratch)
    mislam::DevBuf<float4> tleaf;                        // what the walk reads (NnTreeView): leaf coordinates, ...
    mislam::DevBuf<int> tidx;                            // ... global indices ...
    mislam::DevBuf<float> tboxes6;                       // ... and node boxes, six floats each
    mislam::NnTreeView tree{};
    bool tree_valid = false;
    mislam::DevBuf<float4> gpts;                         // cell grid (nn_grid.h): points sorted by cell
    mislam::DevBuf<unsigned int> gstart, gfill, gscan;   // cell offsets, build cursors, scan scratch
    mislam::DevBuf<unsigned int> gslot_of;               // fixed point -> its slot in gpts
    mislam::DevBuf<unsigned int> grow_occ;               // per cell: which of the 5 x 5 cell rows around it hold a point within reach (NnGridView::row_occ)
    mislam::DevBuf<unsigned char> gnear_tmp;             // build scratch
    mislam::DevBuf<unsigned int> match_slot;             // fused ICP: per moving point, the slot of its current match
    mislam::NnGridView grid{};
    bool grid_valid = false;
    mislam::DevBuf<int> sorder;                          // Morton order of the moving cloud (sorted slot -> caller's index)
    mislam::DevBuf<int> sinv;                            // its inverse (caller's index -> sorted slot), MI_SUM_CPU_SEQUENTIAL only
    mislam::DevBuf<float> resid;                         // per-slot squared residuals, same mode

    // ---- ICP problem currently loaded
    bool icp_loaded = false;
    int n = 0, n_pad = 0;
    int n_global = 0;                                    // moving points over ALL ranks (what rank-independent decisions are taken from)
    int m_total = 0, shard_lo = 0, shard_hi = 0;
    bool source_sharded = false;                         // MI_SHARD_SOURCE in effect: n = this rank's slice, fixed cloud replicated
    bool fused = false;                                  // the grid search carries the O(N) part of the iteration (nn_grid.hip)
    int enqueued_passes = 0;                             // fused iterations enqueued since the load / reset (GridSearchArgs::extend_reach)
    mi_icp_params icp{};
    double load_ms[MI_LOAD_STAGES] = {0};                // mi_icp_load_times

    // ---- CPD workspace (allocated on first use)
    mislam::CpdWorkspace* cpd = nullptr;

    // ---- profiling
    mislam::DevBuf<unsigned long long> nn_stats;         // mi_profile_search_stats counters
    bool nn_stats_on = false;
    int selftest_fail_loads = 0;                         // mi_selftest_fail_loads: the next N index builds fail on purpose (an explicit call of a test, never the environment: ADVICE r05)
    bool profile = false;
    unsigned int prof_mask = 0xffffffffu;                // kernels that get events while profiling (mi_profile_select)
    std::vector<mislam::ProfileSpan> spans;
    std::vector<hipEvent_t> event_pool;
    double prof_ms[MI_KERNEL_COUNT] = {0};
    long long prof_n[MI_KERNEL_COUNT] = {0};

    int prof_begin(int kernel);
    int prof_end();
    int prof_span(int kernel, hipEvent_t* e0, hipEvent_t* e1);   // a span whose events the LAUNCH records (hipExtLaunchKernelGGL); null events if `kernel` is not being timed
    int prof_collect();
};

namespace mislam {

// Every entry point that works on a context opens one: the context's device becomes current and buffers outgrown during the call
// are retired into the context's own list.  On the way out, whatever is on that list is released if the stream happens to be
// drained (the entry points that return host results end in a synchronisation), else it waits for the next one.
struct CtxScope {
    mi_ctx* ctx;
    std::vector<void*>* outer;
    explicit CtxScope(mi_ctx* c);
    ~CtxScope();
    CtxScope(const CtxScope&) = delete;
    CtxScope& operator=(const CtxScope&) = delete;
};
#define MI_ENTER(c)                          \
    MI_HIP(hipSetDevice((c)->device));       \
    mislam::CtxScope mi_ctx_scope_(c)

// the helpers called inside enqueue on `s` and use scratch set `lane` (mi_ctx::work_stream)
struct LaneScope {
    mi_ctx* c;
    LaneScope(mi_ctx* ctx, hipStream_t s, int lane) : c(ctx) { c->lane_stream = s; c->lane = lane; }
    ~LaneScope() { c->lane_stream = nullptr; c->lane = 0; }
};

struct ProfScope {
    mi_ctx* c;
    bool on;
    ProfScope(mi_ctx* ctx, int kernel) : c(ctx), on(ctx->profile && ((ctx->prof_mask >> kernel) & 1u)) { if (on) (void)c->prof_begin(kernel); }
    ~ProfScope() { if (on) (void)c->prof_end(); }
};

struct NnPlan {
    int R, n_chunks, chunk_len;
};
NnPlan plan_nn(const mi_ctx* ctx, int n, int m_local);
size_t target_alloc_len(int m_local);

// Host -> device on the context's stream, through the context's pinned staging buffer for anything sizeable (returns at once;
// the caller's buffer may be reused as soon as this returns).
int host_to_device(mi_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
// Upload an AoS host cloud into SoA device arrays of n_pad entries (tail = copies of the last point).
int upload_soa(mi_ctx* ctx, const float* host_aos, int n, int n_pad, float* x, float* y, float* z, float4* packed);
// Uploads this rank's shard of the fixed cloud (SoA streams for K1 + float4 for gathers); invalidates the box hierarchy.
int upload_target_shard(mi_ctx* ctx, const float* after_xyz, int m_total, bool replicate = false);
// Correspondence search of n moving points (SoA, padded) against the loaded fixed-cloud shard into ctx->keys (K1, K1t or K1g).
int resolve_nn_mode(const mi_ctx* ctx, int nn_mode, int m_local);
int launch_nn(mi_ctx* ctx, const float* sx, const float* sy, const float* sz, int n, int m_local, int index_base, int fma,
              const int* done_flag, int nn_mode);
// In-place all-reduce of a device array over the ranks, on the context's stream: RCCL, or the caller's transport through pinned
// host memory.  No-ops on a single-GPU context.
int allreduce_min_u64(mi_ctx* ctx, unsigned long long* dev_ptr, size_t count);
int allreduce_sum_f64(mi_ctx* ctx, double* dev_ptr, size_t count);

}  // namespace mislam
