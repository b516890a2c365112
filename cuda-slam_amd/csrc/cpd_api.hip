// C ABI, CPD part: rigid CPD driver (replaces CudaCPD, source/cuda-slam/cpdcuda.cu:302-361) and the test-grade
// E-step / M-step / sigma^2 primitives.  The EM loop is enqueued on one stream; the host reads the state block back
// every `sync_every` iterations only to learn whether the device-side stop rule has fired.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "context.h"
#include "cpd_fgt.h"
#include "cpd_kernels.h"

using namespace mislam;

namespace mislam {

// Buffers of one clustered cloud (FgtClusters, cpd_fgt.h)
struct FgtSide {
    DevBuf<float> dist, xc;
    DevBuf<int> indx, memb, off;
    DevBuf<unsigned char> sweep;   // scratch of the grid-wide sweep (large clouds)
    DevBuf<int> picked, replay_state;             // the sweep's choices; the replay's verdict (fgt_replay_kernel)
    DevBuf<unsigned long long> replay_partial;
    int swept_K = 0;            // the fixed cloud only: centres of the sweep dist/indx currently hold (0 = none); see fgt_kcenter_kernel
    int guess_K = 0;            // the moving cloud only: leading entries of `picked` the last E-step's sweep left -- the next one's guess
    int prelaunched = 0;        // > 0: a replay of that guess (this many centres) is on the stream already, behind the last transform
    void release()
    {
        dist.release(); xc.release(); indx.release(); memb.release(); off.release(); sweep.release(); picked.release(); replay_state.release();
        replay_partial.release(); swept_K = 0; guess_K = 0; prelaunched = 0;
    }
};

// Fast-Gauss-Transform E-step workspace ("approximation-type" full / hybrid)
struct FgtWork {
    FgtSide y, a;
    DevBuf<float> By, Ba;                // coefficients: [K][pd][1] (moving cloud as sources), [K][pd][4] (fixed cloud as sources)
    DevBuf<float> By_part, Ba_part;      // big cells (round 5): the partial sums of fgt_model_splits workgroups per cell, [Z][K][pd][W]
    DevBuf<float> kt1, v4;               // transform outputs, per split of the cells: [S][n], [S][4][m]
    DevBuf<unsigned char> sort_temp;
    DevBuf<unsigned char> sort_temp_a;   // the fixed side's own scratch when its clustering runs beside the moving side's (round 5)
    DevBuf<unsigned int> mono;
    DevBuf<float> ck;
    DevBuf<int> hpos;
    int p = 0, pd = 0;
    void release()
    {
        y.release(); a.release(); By.release(); Ba.release(); By_part.release(); Ba_part.release(); kt1.release(); v4.release(); sort_temp.release(); sort_temp_a.release();
        mono.release(); ck.release(); hpos.release(); p = pd = 0;
    }
};

struct CpdWorkspace {
    FgtWork fgt;
    DevBuf<float> ax, ay, az;            // fixed cloud, SoA
    DevBuf<float> den_part, pt1, p1_part, px_part, p1, px;
    DevBuf<float4> xw4;
    DevBuf<double> part_x, part_k, part_init;
    DevBuf<unsigned char> sig_scratch;   // cpd_sigma2_sequential
    // K7t (cpd_trunc.hip): both clouds along a space-filling curve -- the orders (once per load), the fixed cloud's sorted copy and boxes (once),
    // the moving cloud's current positions in curve order and their boxes (every truncated E-step)
    DevBuf<float> t_ax, t_ay, t_az, t_yx, t_yy, t_yz, t_abox, t_ybox, t_bbox;
    DevBuf<int> t_aorder, t_border, t_order_tmp;
    DevBuf<unsigned int> t_codes_in, t_codes_out;
    DevBuf<unsigned char> t_sort;
    DevBuf<float4> t_xw4;
    bool trunc_ready = false;                 // the orders and the fixed cloud's sorted copy belong to the clouds now loaded
    CpdState* d_state = nullptr;
    CpdState* h_state = nullptr;
    int m = 0, n = 0, m_pad = 0, n_pad = 0;   // n = this rank's share of the fixed cloud
    int n_total = 0;                          // |after| over all ranks
    int svd_ieee = 0;                         // the context's MISLAM_SVD_IEEE switch (set where the workspace is fetched)
    int k_chunks = 1, k_chunk_len = 0, x_chunks = 1, x_chunk_len = 0;
    bool sums_fresh = false;                  // the last exact E-step left the M-step's x-sums and k-sums in part_x / part_k
    int sum_rows_x = 0, sum_rows_k = 0;       // ... in this many rows each
    bool replicated = false;                  // multi-rank context, both clouds held WHOLE on every rank (the FGT modes: they cluster whole clouds)
    // Round 6: ... but the E-step's QUERIES are split over the ranks -- the fixed points the first transform is evaluated at
    // [a_lo, a_hi), the moving points of the second [m_lo, m_hi) (mi_shard_range), the truncated E-step's fixed-cloud tiles [t_lo, t_hi) -- and the
    // M-step's 24 sums are all-reduced as in the exact mode.  Clusterings and coefficient tables stay replicated (small, bit-deterministic).
    bool queries_sharded = false;
    int a_lo = 0, a_hi = 0, m_lo = 0, m_hi = 0;
    mi_cpd_params params{};
};

void cpd_workspace_destroy(mi_ctx* c)
{
    CpdWorkspace* w = c->cpd;
    if (!w) return;
    w->ax.release(); w->ay.release(); w->az.release();
    w->den_part.release(); w->pt1.release(); w->p1_part.release(); w->px_part.release(); w->p1.release(); w->px.release();
    w->xw4.release(); w->part_x.release(); w->part_k.release(); w->part_init.release();
    w->t_ax.release(); w->t_ay.release(); w->t_az.release(); w->t_yx.release(); w->t_yy.release(); w->t_yz.release();
    w->t_abox.release(); w->t_ybox.release(); w->t_bbox.release(); w->t_aorder.release(); w->t_border.release(); w->t_order_tmp.release();
    w->t_codes_in.release(); w->t_codes_out.release(); w->t_sort.release(); w->t_xw4.release();
    w->fgt.release();
    if (w->d_state) (void)hipFree(w->d_state);
    if (w->h_state) (void)hipHostFree(w->h_state);
    delete w;
    c->cpd = nullptr;
}

static inline int round_up_i(int v, int g) { return (v + g - 1) / g * g; }

static int cpd_workspace(mi_ctx* c, CpdWorkspace** out)
{
    if (!c->cpd) {
        c->cpd = new CpdWorkspace();
        MI_HIP(hipMalloc((void**)&c->cpd->d_state, sizeof(CpdState)));
        MI_HIP(hipHostMalloc((void**)&c->cpd->h_state, sizeof(CpdState), hipHostMallocDefault));
        memset(c->cpd->h_state, 0, sizeof(CpdState));
    }
    c->cpd->svd_ieee = c->tune.svd_ieee;
    *out = c->cpd;
    return MI_OK;
}

// Chunk the broadcast axis so that the grid has >= ~8 workgroups per CU even for bunny-sized clouds.
static void plan_chunks(const mi_ctx* c, int owners, int owner_r, int stream_len, int* chunks, int* chunk_len)
{
    const int owner_blocks = std::max(1, (owners + 256 * owner_r - 1) / (256 * owner_r));
    int ch = (c->cu_count * 8 + owner_blocks - 1) / owner_blocks;
    ch = std::max(1, std::min(ch, std::min(CPD_MAX_CHUNKS, std::max(1, stream_len / (CPD_T * 8)))));
    *chunk_len = round_up_i((stream_len + ch - 1) / ch, CPD_T);
    *chunks = (stream_len + *chunk_len - 1) / *chunk_len;
}

// Uploads both clouds and sizes every buffer.  y starts as a copy of b (transformedCloud = cloudBefore, :104).
static int cpd_load(mi_ctx* c, CpdWorkspace* w, const float* before_xyz, int m, const float* after_xyz, int n)
{
    c->icp_loaded = false;   // the moving-cloud buffers are shared with the ICP driver
    w->fgt.a.swept_K = 0;    // a new fixed cloud: its clustering starts over
    w->fgt.y.guess_K = 0;    // a new moving cloud: nothing to guess its sweep from
    w->fgt.y.prelaunched = 0;
    w->m = m; w->n = n; w->n_total = n;
    w->trunc_ready = false;
    w->m_pad = round_up_i(m, NN_SRC_PAD);
    w->n_pad = round_up_i(n, NN_SRC_PAD);
    const size_t mp = (size_t)w->m_pad, np = (size_t)w->n_pad;
    MI_TRY(c->bx.reserve(mp)); MI_TRY(c->by.reserve(mp)); MI_TRY(c->bz.reserve(mp));
    MI_TRY(c->cx.reserve(mp)); MI_TRY(c->cy.reserve(mp)); MI_TRY(c->cz.reserve(mp));
    MI_TRY(w->ax.reserve(np)); MI_TRY(w->ay.reserve(np)); MI_TRY(w->az.reserve(np));
    plan_chunks(c, n, 4, m, &w->k_chunks, &w->k_chunk_len);
    plan_chunks(c, m, 2, n, &w->x_chunks, &w->x_chunk_len);
    MI_TRY(w->den_part.reserve((size_t)w->k_chunks * n));
    MI_TRY(w->pt1.reserve(np));
    MI_TRY(w->xw4.reserve(np));
    MI_TRY(w->p1_part.reserve((size_t)w->x_chunks * m));
    MI_TRY(w->px_part.reserve((size_t)w->x_chunks * 3 * m));
    MI_TRY(w->p1.reserve(mp));
    MI_TRY(w->px.reserve(3 * mp));
    MI_TRY(w->part_x.reserve((size_t)std::max(ICP_MAX_PARTIAL_BLOCKS, CPD_TRUNC_MAX_BLOCKS) * CPD_XSUMS));
    MI_TRY(w->part_k.reserve((size_t)std::max(ICP_MAX_PARTIAL_BLOCKS, CPD_TRUNC_MAX_BLOCKS) * CPD_KSUMS));
    MI_TRY(w->part_init.reserve((size_t)ICP_MAX_PARTIAL_BLOCKS * CPD_INIT_SUMS));
    MI_TRY(upload_soa(c, before_xyz, m, w->m_pad, c->bx.p, c->by.p, c->bz.p, nullptr));
    MI_TRY(upload_soa(c, after_xyz, n, w->n_pad, w->ax.p, w->ay.p, w->az.p, nullptr));
    const size_t bytes = sizeof(float) * mp;
    MI_HIP(hipMemcpyAsync(c->cx.p, c->bx.p, bytes, hipMemcpyDeviceToDevice, c->stream));
    MI_HIP(hipMemcpyAsync(c->cy.p, c->by.p, bytes, hipMemcpyDeviceToDevice, c->stream));
    MI_HIP(hipMemcpyAsync(c->cz.p, c->bz.p, bytes, hipMemcpyDeviceToDevice, c->stream));
    return MI_OK;
}

static CpdView cpd_view(mi_ctx* c, CpdWorkspace* w)
{
    CpdView v{};
    v.state = w->d_state;
    v.bx = c->bx.p; v.by = c->by.p; v.bz = c->bz.p;
    v.yx = c->cx.p; v.yy = c->cy.p; v.yz = c->cz.p;
    v.m = w->m;
    v.ax = w->ax.p; v.ay = w->ay.p; v.az = w->az.p;
    v.n = w->n;
    v.den_part = w->den_part.p; v.xw4 = w->xw4.p; v.pt1 = w->pt1.p;
    v.p1_part = w->p1_part.p; v.px_part = w->px_part.p; v.p1 = w->p1.p; v.px = w->px.p;
    v.k_chunks = w->k_chunks; v.k_chunk_len = w->k_chunk_len;
    v.x_chunks = w->x_chunks; v.x_chunk_len = w->x_chunk_len;
    return v;
}

static CpdRules cpd_rules(const CpdWorkspace* w, const mi_cpd_params* p)
{
    CpdRules r{};
    r.eps = p->eps;
    float weight = p->weight;
    if (weight <= 0.0f) weight = 1e-6f;            // coherentpointdrift.cpp:93-96
    if (weight >= 1.0f) weight = 1.0f - 1e-6f;
    r.weight = weight;
    r.tolerance = p->tolerance;
    r.const_scale = p->const_scale;
    r.max_iterations = p->max_iterations;
    r.m = w->m; r.n = w->n_total;
    r.svd_ieee = w->svd_ieee;
    return r;
}

// workgroups of the E-step's post kernels (a quad of lanes per point there: 64 points per workgroup) = rows of M-step partial sums they leave
static int cpd_sum_blocks(int n) { return std::max(1, std::min(ICP_MAX_PARTIAL_BLOCKS, (n + 63) / 64)); }
// ... and of the STAND-ALONE sums kernels (cpd_xsums / cpd_ksums: one lane per point, 256 per workgroup).  ADVICE r04: launched with
// cpd_sum_blocks they left three quarters of their workgroups storing zeros below 32 768 points.  Their fp64 partials are grouped differently
// from the fused ones' (256 against 64 points per row) -- the same sums to ~1e-16 relative, not the same bits.
static int cpd_standalone_sum_blocks(int n) { return std::max(1, std::min(ICP_MAX_PARTIAL_BLOCKS, (n + 255) / 256)); }

static int use_mfma_contraction(const mi_ctx* c) { return c->tune.cpd_mfma; }   // MISLAM_CPD_MFMA, read at context creation

static int cpd_estep_enqueue(mi_ctx* c, CpdWorkspace* w, const CpdView& v)
{
    if (w->params.estep_mode == MI_ESTEP_CPU_SEQUENTIAL && !v.truncate) {
        // parity mode: cpu-slam's summation order (cpd_kernels.hip); the M-step's sums then come from the stand-alone kernels (sums_fresh = false)
        { ProfScope ps(c, MI_KERNEL_CPD_DENOM); MI_HIP(cpd_estep_sequential(v, c->stream)); }
        w->sums_fresh = false;
        return MI_OK;
    }
    // the two post kernels also accumulate the M-step's moments of what they have just produced (two launches less per EM iteration)
    const int nxb = cpd_sum_blocks(w->n), nkb = cpd_sum_blocks(w->m);
    { ProfScope ps(c, MI_KERNEL_CPD_DENOM); MI_HIP(cpd_denominators(v, c->stream)); }
    MI_HIP(cpd_post_denominators(v, c->stream, w->part_x.p, nxb));
    { ProfScope ps(c, MI_KERNEL_CPD_CONTRACT); MI_HIP(cpd_contract(v, use_mfma_contraction(c), c->stream)); }
    MI_HIP(cpd_post_contract(v, c->stream, w->part_k.p, nkb));
    w->sums_fresh = true;
    w->sum_rows_x = nxb; w->sum_rows_k = nkb;
    return MI_OK;
}

// K7t: the hybrid mode's truncated E-step, culled (cpd_trunc.hip).  MISLAM_CPD_TRUNC_CULL=0: round 4's every-pair truncated kernels.
static int cpd_trunc_tiles(int n) { return (n + CPD_TRUNC_TILE - 1) / CPD_TRUNC_TILE; }

static int cpd_trunc_prepare(mi_ctx* c, CpdWorkspace* w, const CpdView& v)
{
    if (w->trunc_ready) return MI_OK;
    const int big = std::max(w->m, w->n);
    MI_TRY(w->t_bbox.reserve(256 * 6 + 6));
    MI_TRY(w->t_codes_in.reserve(big)); MI_TRY(w->t_codes_out.reserve(big)); MI_TRY(w->t_order_tmp.reserve(big));
    MI_TRY(w->t_sort.reserve(std::max<size_t>(tree_sort_temp_bytes(big), 16)));
    MI_TRY(w->t_aorder.reserve(w->n)); MI_TRY(w->t_border.reserve(w->m));
    const size_t na = (size_t)cpd_trunc_tiles(w->n) * CPD_TRUNC_TILE, ny = (size_t)cpd_trunc_tiles(w->m) * CPD_TRUNC_TILE;
    MI_TRY(w->t_ax.reserve(na)); MI_TRY(w->t_ay.reserve(na)); MI_TRY(w->t_az.reserve(na)); MI_TRY(w->t_xw4.reserve(na));
    MI_TRY(w->t_yx.reserve(ny)); MI_TRY(w->t_yy.reserve(ny)); MI_TRY(w->t_yz.reserve(ny));
    constexpr int per_tile = 1 + CPD_TRUNC_TILE / CPD_TRUNC_GROUP;          // a tile's box + its groups' boxes
    MI_TRY(w->t_abox.reserve(6 * (size_t)per_tile * cpd_trunc_tiles(w->n)));
    MI_TRY(w->t_ybox.reserve(6 * (size_t)per_tile * cpd_trunc_tiles(w->m)));
    // the fixed cloud along its curve; the moving cloud along the curve of its ORIGINAL points (a similarity transform keeps neighbours together)
    MortonArgs ma{};
    ma.bbox_partials = w->t_bbox.p; ma.bbox = w->t_bbox.p + 256 * 6;
    ma.codes_in = w->t_codes_in.p; ma.codes_out = w->t_codes_out.p; ma.order_in = w->t_order_tmp.p;
    ma.sort_temp = w->t_sort.p; ma.sort_temp_bytes = w->t_sort.cap;
    ma.x = v.ax; ma.y = v.ay; ma.z = v.az; ma.m = w->n; ma.order_out = w->t_aorder.p;
    MI_HIP(morton_order(ma, c->stream));
    ma.x = v.bx; ma.y = v.by; ma.z = v.bz; ma.m = w->m; ma.order_out = w->t_border.p;
    MI_HIP(morton_order(ma, c->stream));
    MI_HIP(cpd_trunc_gather(v.ax, v.ay, v.az, w->t_aorder.p, w->n, w->t_ax.p, w->t_ay.p, w->t_az.p, w->t_abox.p,
                            w->t_abox.p + 6 * (size_t)cpd_trunc_tiles(w->n), nullptr, c->stream));
    w->trunc_ready = true;
    return MI_OK;
}

static int cpd_estep_trunc_enqueue(mi_ctx* c, CpdWorkspace* w, const CpdView& v)
{
    MI_TRY(cpd_trunc_prepare(c, w, v));
    CpdTruncView t{};
    t.state = v.state;
    t.ax = w->t_ax.p; t.ay = w->t_ay.p; t.az = w->t_az.p;
    t.abox = w->t_abox.p; t.agroup = w->t_abox.p + 6 * (size_t)cpd_trunc_tiles(w->n);
    t.a_order = w->t_aorder.p; t.n = w->n;
    t.yx = w->t_yx.p; t.yy = w->t_yy.p; t.yz = w->t_yz.p;
    t.ybox = w->t_ybox.p; t.ygroup = w->t_ybox.p + 6 * (size_t)cpd_trunc_tiles(w->m);
    t.b_order = w->t_border.p; t.m = w->m;
    t.bx = v.bx; t.by = v.by; t.bz = v.bz;
    t.xw4 = w->t_xw4.p; t.xw4_caller = v.xw4; t.pt1 = v.pt1; t.p1 = v.p1; t.px = v.px;
    t.trunc_log = v.trunc_log;
    // a multi-rank context: this rank's tiles of the FIXED cloud (curve order) for the denominators, its tiles of the MOVING cloud for the contraction, each
    // against the whole other cloud -- per-point values bit for bit the single-GPU run's; the M-step's sums are added over the ranks
    t.a_tile_lo = 0; t.a_tile_hi = cpd_trunc_tiles(w->n); t.y_tile_lo = 0; t.y_tile_hi = cpd_trunc_tiles(w->m);
    if (w->queries_sharded) {
        (void)mi_shard_range(cpd_trunc_tiles(w->n), c->rank, c->world, &t.a_tile_lo, &t.a_tile_hi);
        (void)mi_shard_range(cpd_trunc_tiles(w->m), c->rank, c->world, &t.y_tile_lo, &t.y_tile_hi);
    }
    const int nxb = std::max(1, std::min(CPD_TRUNC_MAX_BLOCKS, t.a_tile_hi - t.a_tile_lo)), nkb = std::max(1, std::min(CPD_TRUNC_MAX_BLOCKS, t.y_tile_hi - t.y_tile_lo));
    // the moving cloud's current positions in curve order + this E-step's boxes
    MI_HIP(cpd_trunc_gather(v.yx, v.yy, v.yz, w->t_border.p, w->m, w->t_yx.p, w->t_yy.p, w->t_yz.p, w->t_ybox.p,
                            w->t_ybox.p + 6 * (size_t)cpd_trunc_tiles(w->m), v.state, c->stream));
    { ProfScope ps(c, MI_KERNEL_CPD_DENOM); MI_HIP(cpd_trunc_denominators(t, w->part_x.p, nxb, c->stream)); }
    if (w->queries_sharded) {
        // the contraction's operand (w x, w y, w z, w per fixed point, curve order): every rank has written its own tiles' points; an unsigned MINIMUM against
        // all-ones hands every rank all of them, bit for bit (as for the FGT E-step's weights below)
        const size_t p_lo = (size_t)t.a_tile_lo * CPD_TRUNC_TILE, p_hi = std::min((size_t)t.a_tile_hi * CPD_TRUNC_TILE, (size_t)w->n);
        if (p_lo > 0) MI_HIP(hipMemsetAsync(w->t_xw4.p, 0xff, sizeof(float4) * p_lo, c->stream));
        if (p_hi < (size_t)w->n) MI_HIP(hipMemsetAsync(w->t_xw4.p + p_hi, 0xff, sizeof(float4) * ((size_t)w->n - p_hi), c->stream));
        MI_TRY(allreduce_min_u64(c, reinterpret_cast<unsigned long long*>(w->t_xw4.p), 2 * (size_t)w->n));
    }
    { ProfScope ps(c, MI_KERNEL_CPD_CONTRACT); MI_HIP(cpd_trunc_contract(t, w->part_k.p, nkb, c->stream)); }
    w->sums_fresh = true;
    w->sum_rows_x = nxb; w->sum_rows_k = nkb;
    return MI_OK;
}

static int cpd_mstep_enqueue(mi_ctx* c, CpdWorkspace* w, const CpdView& v, const CpdRules& rules, int update_loop_state)
{
    int nxb = cpd_standalone_sum_blocks(w->n), nkb = cpd_standalone_sum_blocks(w->m);
    ProfScope ps(c, MI_KERNEL_CPD_MSTEP);
    if (!w->sums_fresh) {
        MI_HIP(cpd_xsums(v, w->part_x.p, nxb, c->stream));
        MI_HIP(cpd_ksums(v, w->part_k.p, nkb, c->stream));
    } else { nxb = w->sum_rows_x; nkb = w->sum_rows_k; }     // (the rows the E-step's own kernels left)
    w->sums_fresh = false;
    if (!c->distributed() || (w->replicated && !w->queries_sharded)) {
        MI_HIP(cpd_solve(w->d_state, w->part_x.p, nxb, w->part_k.p, nkb, rules, update_loop_state, c->stream));
        return MI_OK;
    }
    // Fixed cloud sharded over the ranks (C2): the x-sums cover this rank's fixed points, the k-sums are linear
    // in this rank's share of P1/PX -- ONE all-reduce of the 24 doubles gives every rank the full moments; the per-point P1/PX
    // never travel.  Every rank then runs the same solve.
    static_assert(offsetof(CpdState, ks) == offsetof(CpdState, xs) + sizeof(double) * CPD_XSUMS, "xs and ks must be contiguous");
    MI_HIP(cpd_reduce_sums(w->d_state, w->part_x.p, nxb, w->part_k.p, nkb, c->stream));
    MI_TRY(allreduce_sum_f64(c, w->d_state->xs, (size_t)(CPD_XSUMS + CPD_KSUMS)));
    MI_HIP(cpd_solve(w->d_state, nullptr, 0, nullptr, 0, rules, update_loop_state, c->stream));
    return MI_OK;
}

static int cpd_fetch(mi_ctx* c, CpdWorkspace* w)
{
    MI_HIP(hipMemcpyAsync(w->h_state, w->d_state, sizeof(CpdState), hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipStreamSynchronize(c->stream));
    return MI_OK;
}

static int cpd_init(mi_ctx* c, CpdWorkspace* w, const CpdView& v, const CpdRules& rules, float sigma2_override, int sigma2_mode)
{
    const int nb = icp_reduce_blocks(std::max(w->m, w->n));
    MI_HIP(cpd_init_sums(v, w->part_init.p, nb, c->stream));
    const int seq = sigma2_mode == MI_SIGMA2_CPU_SEQUENTIAL && !(sigma2_override > 0.f);
    if (seq) {
        if (c->distributed() && !w->replicated) { set_error("CPD: MI_SIGMA2_CPU_SEQUENTIAL needs a single-GPU context (one running sum over all pairs)"); return MI_ERR_INVALID_ARG; }
        MI_TRY(w->sig_scratch.reserve(cpd_sigma2_scratch_bytes()));
        MI_HIP(cpd_sigma2_sequential(v, c->stream, w->sig_scratch.p, (int*)c->h_scratch));
    }
    if (!c->distributed() || w->replicated) {
        MI_HIP(cpd_init_state(w->d_state, w->part_init.p, nb, rules, sigma2_override, seq, c->stream));
        return MI_OK;
    }
    // sharded fixed cloud: its four sums (init[0..3]) are added over the ranks; the moving cloud's (init[4..7]) are replicated
    MI_HIP(cpd_reduce_init(w->d_state, w->part_init.p, nb, c->stream));
    MI_TRY(allreduce_sum_f64(c, w->d_state->init, 4));
    MI_HIP(cpd_init_state(w->d_state, nullptr, 0, rules, sigma2_override, 0, c->stream));
    return MI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Fast Gauss Transform E-step (ComputePMatrixWithFGT, common/cpdutils.cpp:19-77)
// ---------------------------------------------------------------------------------------------------------------
void fgt_build_tables(int p, std::vector<unsigned int>& mono, std::vector<float>& ck, std::vector<int>& hpos)
{
    const int pd = p * (p + 1) * (p + 2) / 6;                      // nchoosek(p + 2, 3), fgt.cpp:69
    mono.assign(pd, 0u); ck.assign(pd, 0.f); hpos.assign(pd, 0);
    // slot of exponents (a, b, c) in the reference's graded order: all of degree d-1 first, then x-power descending, then y-power
    auto slot = [](int a, int b, int c) { const int d = a + b + c, r = d - a; return d * (d + 1) * (d + 2) / 6 + r * (r + 1) / 2 + (r - b); };
    for (int d = 0; d < p; d++)
        for (int a = d; a >= 0; a--)
            for (int b = d - a; b >= 0; b--) {
                const int c = d - a - b, t = slot(a, b, c);
                mono[t] = (unsigned)a | ((unsigned)b << 8) | ((unsigned)c << 16);
                // 2^|alpha| / alpha!, rounded the way ComputeC_k walks the recursion: z steps, then y steps, then x steps, each step
                // C = float(2.0 * C) followed by C = float(C / (double)q), q = the running exponent (fgt.cpp:236-238)
                float C = 1.0f;
                const int steps[3] = {c, b, a};
                for (int axis = 0; axis < 3; axis++)
                    for (int q = 1; q <= steps[axis]; q++) { C = (float)(2.0 * C); C = (float)(C / (double)q); }
                ck[t] = C;
            }
    int h = 0;                                                     // Horner traversal of fgt_predict_kernel
    for (int a = p - 1; a >= 0; a--)
        for (int b = p - 1 - a; b >= 0; b--)
            for (int c = p - 1 - a - b; c >= 0; c--) hpos[slot(a, b, c)] = h++;
}

static int fgt_tables(mi_ctx* c, FgtWork* f, int p, FgtTables* out)
{
    if (p < 1 || p > FGT_MAX_ORDER) { set_error("FGT: order of truncation %d outside [1, %d]", p, FGT_MAX_ORDER); return MI_ERR_INVALID_ARG; }
    if (f->p != p) {
        std::vector<unsigned int> mono; std::vector<float> ck; std::vector<int> hpos;
        fgt_build_tables(p, mono, ck, hpos);
        const size_t pd = mono.size();
        MI_TRY(f->mono.reserve(pd)); MI_TRY(f->ck.reserve(pd)); MI_TRY(f->hpos.reserve(pd));
        MI_HIP(hipMemcpyAsync(f->mono.p, mono.data(), pd * sizeof(unsigned int), hipMemcpyHostToDevice, c->stream));
        MI_HIP(hipMemcpyAsync(f->ck.p, ck.data(), pd * sizeof(float), hipMemcpyHostToDevice, c->stream));
        MI_HIP(hipMemcpyAsync(f->hpos.p, hpos.data(), pd * sizeof(int), hipMemcpyHostToDevice, c->stream));
        MI_HIP(hipStreamSynchronize(c->stream));                   // the staging vectors die here
        f->p = p; f->pd = (int)pd;
    }
    out->mono = f->mono.p; out->ck = f->ck.p; out->hpos = f->hpos.p; out->p = f->p; out->pd = f->pd;
    return MI_OK;
}

static int fgt_side(mi_ctx* c, FgtWork* f, FgtSide* sd, const float* x, const float* y, const float* z, int n, int K, FgtClusters* out)
{
    MI_TRY(sd->dist.reserve(n)); MI_TRY(sd->indx.reserve(n)); MI_TRY(sd->memb.reserve(n));
    MI_TRY(sd->off.reserve((size_t)K + 1)); MI_TRY(sd->xc.reserve(3 * (size_t)K));
    MI_TRY(f->sort_temp.reserve(std::max<size_t>(fgt_sort_temp_bytes(n), 16)));
    if (n > 16 * 1024) MI_TRY(sd->sweep.reserve(FGT_SWEEP_SCRATCH_BYTES));        // (beyond what one workgroup's registers hold: the cooperative / grid-wide sweeps)
    out->sweep_scratch = n > 16 * 1024 ? sd->sweep.p : nullptr;
    out->coop_sweep = c->tune.fgt_coop_sweep;
    // (picked: never shrunk below what a guess still needs -- reserve() keeps the contents when the capacity suffices, and K only grows
    // within a registration; a reallocation loses the guess, so it is dropped with it)
    if (sd->picked.cap < (size_t)K) sd->guess_K = 0;
    MI_TRY(sd->picked.reserve(std::max<size_t>((size_t)K, 1024))); MI_TRY(sd->replay_state.reserve(1));
    out->picked = sd->picked.p; out->guess = 0; out->replay_partial = nullptr; out->replay_state = sd->replay_state.p;
    out->x = x; out->y = y; out->z = z; out->n = n; out->K = K; out->k_done = 0;
    out->dist = sd->dist.p; out->indx = sd->indx.p;
    out->memb = sd->memb.p; out->off = sd->off.p; out->xc = sd->xc.p;
    return MI_OK;
}

// cl->picked[0 .. guess) holds a guess: size the replay's scratch and arm it (guess < 2: nothing to replay)
static int fgt_arm_replay(mi_ctx* c, FgtSide* sd, FgtClusters* cl, int guess)
{
    (void)c;
    const int lim = fgt_replay_limit(guess, cl->K);
    if (lim < 1) return MI_OK;
    MI_TRY(sd->replay_partial.reserve((size_t)lim * (size_t)fgt_replay_waves(cl->n)));
    cl->guess = guess; cl->replay_partial = sd->replay_partial.p;
    return MI_OK;
}

// Behind the transform that ends an EM iteration, before the host has sigma^2 (hence K) of the next one: the moving cloud's sweep replayed
// from this iteration's choices.  The replay needs the transformed cloud and the guess, not K; K only grows while the FGT is in use, so the
// next E-step finds exactly the replay it would have launched -- 20 us of work inside the ~25 us the device used to wait for the host.
static int cpd_fgt_prelaunch(mi_ctx* c, CpdWorkspace* w, const CpdView& v)
{
    FgtWork* f = &w->fgt;
    f->y.prelaunched = 0;
    if (c->tune.fgt_replay == 0 || f->y.guess_K < 2) return MI_OK;
    FgtClusters cy{};
    MI_TRY(fgt_side(c, f, &f->y, v.yx, v.yy, v.yz, w->m, f->y.guess_K, &cy));
    if (f->y.guess_K < 2) return MI_OK;                 // (fgt_side dropped the guess with a reallocated buffer)
    MI_TRY(fgt_arm_replay(c, &f->y, &cy, f->y.guess_K));
    MI_HIP(fgt_replay_prelaunch(cy, c->stream));
    f->y.prelaunched = fgt_replay_limit(cy.guess, cy.K);
    return MI_OK;
}

// K of the transform: cpdutils.cpp:36
static int fgt_cluster_count(int m, int n, float sigma2, float sigma2_init)
{
    return (int)std::round(std::min({(float)n, (float)m, 50.0f + sigma2_init / sigma2}));
}

// ndi: cpdutils.cpp:45 -- pow and the numerator in double, the denominator a float product, narrowed to float
static float fgt_ndi(float sigma2, float weight, int m, int n)
{
    return (float)((std::pow(2 * 3.14159265358979323846 * sigma2, (double)(3.f * 0.5f)) * weight * m) / (double)((1 - weight) * n));
}

// with_sums: the two post kernels also leave the M-step's x-sums / k-sums (the driver's E-steps; a stand-alone E-step reads P1 / Pt1 / PX back instead)
static int cpd_estep_fgt_enqueue(mi_ctx* c, CpdWorkspace* w, const CpdView& v, float weight, float sigma2, float sigma2_init,
                                 float ratio_of_far_field, int order, bool with_sums = false)
{
    ProfScope ps(c, MI_KERNEL_CPD_FGT);
    w->sums_fresh = false;               // P1 / Pt1 / PX come from the transform below, not from the exact E-step's post kernels
    FgtWork* f = &w->fgt;
    if (w->m < 2 || w->n < 2) { set_error("FGT E-step needs at least 2 points per cloud (m=%d, n=%d)", w->m, w->n); return MI_ERR_INVALID_ARG; }
    if (!(sigma2 > 0.f)) { set_error("FGT E-step: sigma2 must be positive"); return MI_ERR_INVALID_ARG; }
    FgtTables t{};
    MI_TRY(fgt_tables(c, f, order, &t));
    const int K = fgt_cluster_count(w->m, w->n, sigma2, sigma2_init);
    if (K < 1 || K > FGT_MAX_CLUSTERS) { set_error("FGT E-step: %d cells outside [1, %d]", K, FGT_MAX_CLUSTERS); return MI_ERR_INVALID_ARG; }
    const float hsigma = std::sqrt(2.0f * sigma2);                 // cpdutils.cpp:31
    const float ndi = fgt_ndi(sigma2, weight, w->m, w->n);
    FgtClusters cy{}, ca{};
    MI_TRY(fgt_side(c, f, &f->y, v.yx, v.yy, v.yz, w->m, K, &cy));
    MI_TRY(fgt_side(c, f, &f->a, v.ax, v.ay, v.az, w->n, K, &ca));
    MI_TRY(f->By.reserve((size_t)K * t.pd)); MI_TRY(f->Ba.reserve(4 * (size_t)K * t.pd));
    // (big cells: a cell's members over several workgroups -- unless the cloud is one whose cells list their own members, which is one workgroup per cell;
    // decided from n, K and pd ALONE: ADVICE r05 found the fixed side taking the listing path on a re-clustering E-step and the split path on the others,
    // two orders of the same sums, for clouds of at most 32 768 points in cells of 1 024 and more)
    const auto splits = [&](int n_side) {
        if (c->tune.fgt_model_splits == 0) return 1;
        if (c->tune.fgt_lists_in_model != 0 && fgt_lists_rule(n_side, K)) return 1;
        return fgt_model_splits(n_side, K, t.pd);
    };
    const int Zy = splits(w->m), Za = splits(w->n);
    if (Zy > 1) MI_TRY(f->By_part.reserve((size_t)Zy * K * t.pd));
    if (Za > 1) MI_TRY(f->Ba_part.reserve(4 * (size_t)Za * K * t.pd));
    const int Sa = fgt_predict_splits(w->n, K), Sy = fgt_predict_splits(w->m, K);
    MI_TRY(f->kt1.reserve((size_t)Sa * w->n)); MI_TRY(f->v4.reserve(4 * (size_t)Sy * w->m));
    const size_t temp = f->sort_temp.cap;
    // Round 5: the FIXED cloud's clustering (the additional centres of a larger K, its member lists: ~18 us of small launches) depends on nothing the
    // moving side computes -- it runs on the context's auxiliary stream, with its own scratch, while `stream` clusters the moving cloud, builds its
    // model and evaluates it; `stream` waits for it just before the fixed side's model.  (With profiling on everything stays on `stream`, one after
    // the other.)  The fixed cloud does not move: the same K needs no new clustering at all, a larger K only the additional centres
    // (a context created under MISLAM_FGT_RESUME=0 re-clusters from scratch every time; the results must not change by a bit -- tests/test_gpu_fgt.py)
    const bool resume = c->tune.fgt_resume != 0;
    const bool recluster = !resume || f->a.swept_K != K;
    const bool beside = recluster && !c->profile && c->aux != nullptr && c->tune.fgt_two_streams != 0;
    struct AuxJoin {                       // every way out joins the auxiliary stream into `stream`
        mi_ctx* c; bool armed;
        ~AuxJoin() { if (armed && hipEventRecord(c->aux_event[1], c->aux) == hipSuccess) (void)hipStreamWaitEvent(c->stream, c->aux_event[1], 0); }
    } aux_join{c, false};
    if (recluster) {
        ca.k_done = resume && f->a.swept_K < K ? f->a.swept_K : 0;
        ca.centers_in_model = 1;
        ca.lists_in_model = c->tune.fgt_lists_in_model;
        if (beside) {
            MI_TRY(f->sort_temp_a.reserve(std::max<size_t>(fgt_sort_temp_bytes(w->n), 16)));
            MI_HIP(hipEventRecord(c->aux_event[0], c->stream));                 // behind the last readers of the fixed side's lists
            MI_HIP(hipStreamWaitEvent(c->aux, c->aux_event[0], 0));
            aux_join.armed = true;
            MI_HIP(fgt_cluster(ca, f->sort_temp_a.p, f->sort_temp_a.cap, c->aux));
        }
    }
    // Kt1 = K^T 1: sources = moving cloud, unit weights, queried at the fixed cloud   (cpdutils.cpp:42-43)
    // (the moving cloud's sweep: last E-step's choices replayed and checked in parallel, MISLAM_FGT_REPLAY=0: swept step by step every time)
    MI_TRY(fgt_arm_replay(c, &f->y, &cy, c->tune.fgt_replay != 0 ? f->y.guess_K : 0));
    // (prelaunched: the same replay went onto the stream behind the last transform, cpd_fgt_prelaunch below -- K has not shrunk below the guess)
    cy.replay_done = f->y.prelaunched > 0 && f->y.prelaunched == fgt_replay_limit(cy.guess, cy.K) ? 1 : 0;
    f->y.prelaunched = 0;
    cy.centers_in_model = 1;                 // (round 5: the cell means come out of the model kernel -- one launch less per side)
    cy.lists_in_model = c->tune.fgt_lists_in_model;      // (... and so do the member lists, for clouds of at most 32 768 points: three launches less per side)
    MI_HIP(fgt_cluster(cy, f->sort_temp.p, temp, c->stream));
    f->y.guess_K = K;
    MI_HIP(fgt_model(cy, nullptr, hsigma, t, f->By.p, c->stream, true, Zy > 1 ? f->By_part.p : nullptr, Zy));
    // The two evaluation kernels carry the E-step's arithmetic and each query is independent of every other: on a multi-rank
    // context a rank evaluates ITS queries only -- with the whole cloud's number of cell splits (Sa, Sy), so that a point's partial sums are grouped as on
    // one GPU and its value is the single-GPU run's to the last bit.
    const bool sharded = w->queries_sharded && with_sums;
    const int a0 = sharded ? w->a_lo : 0, an = sharded ? w->a_hi - w->a_lo : w->n;
    const int m0 = sharded ? w->m_lo : 0, mn = sharded ? w->m_hi - w->m_lo : w->m;
    MI_HIP(fgt_predict(v.ax + a0, v.ay + a0, v.az + a0, an, cy.xc, f->By.p, K, 1, hsigma, ratio_of_far_field, t, Sa, f->kt1.p, c->stream));
    const int nxb = cpd_standalone_sum_blocks(an), nkb = cpd_standalone_sum_blocks(mn);
    MI_HIP(fgt_post_kt1(f->kt1.p, Sa, v.ax + a0, v.ay + a0, v.az + a0, an, ndi, v.pt1 + a0, v.xw4 + a0, c->stream, with_sums ? w->part_x.p : nullptr, nxb));
    if (sharded) {
        // The fixed side's model build weighs EVERY fixed point by its 1/den, x/den (xw4): each rank has written its own range; the others' arrive through
        // the transport.  An element-wise unsigned MINIMUM against all-ones moves any bit pattern unchanged (a sum would not: NaN payloads, -0), so
        // every rank ends up with the bits the single-GPU run has, and builds the same table from them.
        if (a0 > 0) MI_HIP(hipMemsetAsync(v.xw4, 0xff, sizeof(float4) * (size_t)a0, c->stream));
        if (w->a_hi < w->n) MI_HIP(hipMemsetAsync(v.xw4 + w->a_hi, 0xff, sizeof(float4) * (size_t)(w->n - w->a_hi), c->stream));
        MI_TRY(allreduce_min_u64(c, reinterpret_cast<unsigned long long*>(v.xw4), 2 * (size_t)w->n));
    }
    // P1 and PX: sources = fixed cloud weighted by 1/den and x/den, queried at the moving cloud   (:54-66; the reference
    // clusters the fixed cloud four times with the same result -- once is enough)
    if (recluster) {
        if (beside) {
            MI_HIP(hipEventRecord(c->aux_event[1], c->aux));
            MI_HIP(hipStreamWaitEvent(c->stream, c->aux_event[1], 0));
            aux_join.armed = false;
        } else
            MI_HIP(fgt_cluster(ca, f->sort_temp.p, temp, c->stream));
        f->a.swept_K = K;
    }
    MI_HIP(fgt_model(ca, v.xw4, hsigma, t, f->Ba.p, c->stream, recluster, Za > 1 ? f->Ba_part.p : nullptr, Za));      // (an unchanged clustering keeps its means)
    MI_HIP(fgt_predict(v.yx + m0, v.yy + m0, v.yz + m0, mn, ca.xc, f->Ba.p, K, 4, hsigma, ratio_of_far_field, t, Sy, f->v4.p, c->stream));
    MI_HIP(fgt_post_px(f->v4.p, Sy, mn, v.p1 + m0, v.px + 3 * (size_t)m0, c->stream, with_sums ? w->part_k.p : nullptr, nkb, v.bx + m0, v.by + m0, v.bz + m0));
    if (with_sums) { w->sums_fresh = true; w->sum_rows_x = nxb; w->sum_rows_k = nkb; }
    return MI_OK;
}

}  // namespace mislam

extern "C" void mi_cpd_params_default(mi_cpd_params* p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->eps = 1e-3f;            // "convergence-epsilon"  configparser.cpp:244
    p->weight = 0.3f;          // "cpd-weight"           configparser.cpp:238
    p->const_scale = 0;        // "cpd-const-scale"      configparser.cpp:240
    p->max_iterations = -1;    // gpumain.cpp:14 -- NB: runs no iteration, exactly like the reference
    p->tolerance = 1e-3f;      // "cpd-tolerance"        configparser.cpp:242
    p->sigma2_init = 0.f;
    p->sync_every = 0;
    p->verbose = 0;
    p->approximation = MI_CPD_APPROX_NONE;   // the exact P; the reference's parser default is hybrid (configparser.cpp:221-230)
    p->fgt_ratio_of_far_field = 10.0f;       // "fgt-ratio-of-far-field"   configparser.cpp:263
    p->fgt_order_of_truncation = 8;          // "fgt-order-of-truncation"  configparser.cpp:264
}

static int cpd_check(mi_ctx* c, const float* b, int m, const float* a, int n, bool allow_sharded = false)
{
    if (!c) { set_error("CPD: null context"); return MI_ERR_INVALID_ARG; }
    if (!b || !a || m <= 0 || n <= 0) { set_error("CPD: empty or null cloud (m=%d, n=%d)", m, n); return MI_ERR_INVALID_ARG; }
    if (c->world != 1 && !allow_sharded) { set_error("CPD primitives run on single-GPU contexts (mi_cpd_register shards)"); return MI_ERR_STATE; }
    return MI_OK;
}

extern "C" int mi_cpd_register(mi_ctx* c, const float* before_xyz, int m_before, const float* after_xyz, int n_after,
                               const mi_cpd_params* params, float out_sR_t[16], float* out_scale, int* iterations, float* error)
{
    MI_TRY(cpd_check(c, before_xyz, m_before, after_xyz, n_after, true));
    if (!params || !out_sR_t || !iterations || !error) { set_error("mi_cpd_register: null argument"); return MI_ERR_INVALID_ARG; }
    if (n_after < c->world) { set_error("mi_cpd_register: %d fixed points cannot be split over %d ranks", n_after, c->world); return MI_ERR_INVALID_ARG; }
    if (params->approximation < MI_CPD_APPROX_NONE || params->approximation > MI_CPD_APPROX_HYBRID) {
        set_error("mi_cpd_register: unknown approximation %d", params->approximation);
        return MI_ERR_INVALID_ARG;
    }
    if (params->approximation != MI_CPD_APPROX_NONE &&
        (params->fgt_order_of_truncation < 1 || params->fgt_order_of_truncation > FGT_MAX_ORDER || m_before < 2 || n_after < 2)) {
        set_error("mi_cpd_register: the FGT modes need 1 <= order of truncation <= %d and >= 2 points per cloud", FGT_MAX_ORDER);
        return MI_ERR_INVALID_ARG;
    }
    if (params->estep_mode != MI_ESTEP_DEFAULT && params->estep_mode != MI_ESTEP_CPU_SEQUENTIAL) { set_error("mi_cpd_register: bad estep_mode %d", params->estep_mode); return MI_ERR_INVALID_ARG; }
    if (params->estep_mode == MI_ESTEP_CPU_SEQUENTIAL && (c->world > 1 || params->approximation != MI_CPD_APPROX_NONE)) {
        set_error("mi_cpd_register: MI_ESTEP_CPU_SEQUENTIAL is a single-GPU parity mode of the exact E-step");
        return MI_ERR_INVALID_ARG;
    }
    MI_ENTER(c);
    CpdWorkspace* w = nullptr;
    MI_TRY(cpd_workspace(c, &w));
    w->params = *params;
    // every rank is handed both clouds whole (as mi_icp_register) and keeps fixed points [lo, hi) -- mi_shard_range.
    // The FGT modes ("full", and "hybrid" -- the reference parser's default, configparser.cpp:217) cluster the WHOLE fixed cloud and
    // their E-step is O((N + M) K) -- milliseconds where the exact one takes seconds -- so on a multi-rank context they run
    // REPLICATED: every rank keeps both clouds whole and does the same arithmetic in the same order (the same bits everywhere),
    // no collective at all.  The exact P shards.
    int lo = 0, hi = n_after;
    w->replicated = c->world > 1 && params->approximation != MI_CPD_APPROX_NONE;
    w->queries_sharded = w->replicated && c->tune.fgt_shard_queries != 0 && cpd_trunc_tiles(m_before) >= c->world && cpd_trunc_tiles(n_after) >= c->world;
    if (w->queries_sharded) {
        (void)mi_shard_range(n_after, c->rank, c->world, &w->a_lo, &w->a_hi);
        (void)mi_shard_range(m_before, c->rank, c->world, &w->m_lo, &w->m_hi);
    }
    if (c->world > 1 && !w->replicated) (void)mi_shard_range(n_after, c->rank, c->world, &lo, &hi);
    MI_TRY(cpd_load(c, w, before_xyz, m_before, after_xyz + 3 * (size_t)lo, hi - lo));
    w->n_total = n_after;
    const CpdView v = cpd_view(c, w);
    const CpdRules rules = cpd_rules(w, params);
    if (params->sigma2_mode != MI_SIGMA2_EXACT && params->sigma2_mode != MI_SIGMA2_CPU_SEQUENTIAL) { set_error("mi_cpd_register: bad sigma2_mode %d", params->sigma2_mode); return MI_ERR_INVALID_ARG; }
    MI_TRY(cpd_init(c, w, v, rules, params->sigma2_init, params->sigma2_mode));
    MI_TRY(cpd_fetch(c, w));
    int batch = params->sync_every;
    if (batch <= 0) {
        // enough iterations per host check that the check (a 160-byte read-back: ~35 us of idle device) stays a few per cent of them, few
        // enough that what is enqueued past the stopping iteration (kernels that return at once, ~15 us per iteration) stays small: about a
        // millisecond of E-steps.  From the GLOBAL sizes (every rank of a multi-GPU context picks the same batch): the two passes over the
        // pair space run at ~2.5e12 pairs/s on one GPU, the small kernels and gaps take ~60 us per iteration
        const double pairs = (double)m_before * (double)n_after;
        const double est_s = 2.0 * pairs / 2.5e12 / (double)(w->replicated ? 1 : c->world) + 6e-5;
        batch = std::max(1, std::min(8, (int)(1.2e-3 / est_s)));
    }
    if (params->approximation != MI_CPD_APPROX_NONE) batch = 1;   // the E-step's shape depends on sigma^2: host-stepped
    // exact P in batches: a host check only peeks (state copy behind the batch, ONE iteration of the next batch behind the copy, the host
    // waits for the copy alone) -- the device does not idle for the ~26 us a drained stream and a fresh enqueue cost (mi_icp_run does the same)
    const bool pipelined = batch > 1 && c->peek_event != nullptr && c->tune.icp_pipeline != 0 && !c->profile;
    bool ahead = false;
    while (!w->h_state->done) {
        for (int b = ahead ? 1 : 0; b < batch; b++) {
            if (params->approximation == MI_CPD_APPROX_NONE) {
                MI_TRY(cpd_estep_enqueue(c, w, v));
            } else {
                // ComputePMatrixFast, coherentpointdrift.cpp:141-167 (the comparisons are double there: 0.05 and 0.015 are doubles)
                float sigma2 = w->h_state->sigma2;
                const float sigma2_init = w->h_state->sigma2_init;
                bool fgt = true;
                if (params->approximation == MI_CPD_APPROX_FULL) {
                    if ((double)sigma2 < 0.05) {
                        sigma2 = (float)0.05;
                        MI_HIP(cpd_set_sigma2(w->d_state, sigma2, c->stream));
                    }
                } else {
                    fgt = (double)sigma2 > 0.015 * (double)sigma2_init;
                }
                if (fgt) {
                    MI_TRY(cpd_estep_fgt_enqueue(c, w, v, rules.weight, sigma2, sigma2_init, params->fgt_ratio_of_far_field,
                                                 params->fgt_order_of_truncation, true));
                } else {
                    CpdView vt = v;
                    vt.truncate = 1;
                    vt.trunc_log = std::log(1e-3f);                // ComputePMatrix(..., true, 1e-3f), :166 / :182-183
                    // (MISLAM_CPD_TRUNC_CULL=0 -- round 4's every-pair truncated kernels -- has no sharded form: a sharded registration takes the culled ones)
                    if (c->tune.cpd_trunc_cull != 0 || w->queries_sharded) MI_TRY(cpd_estep_trunc_enqueue(c, w, vt));
                    else MI_TRY(cpd_estep_enqueue(c, w, vt));
                }
            }
            MI_TRY(cpd_mstep_enqueue(c, w, v, rules, 1));
            MI_HIP(cpd_transform(v, w->m_pad, c->stream));
        }
        if (params->approximation != MI_CPD_APPROX_NONE && c->peek_event != nullptr) {
            // the state copy first, the replay BEHIND it: the host has sigma^2 while the device replays
            MI_HIP(hipMemcpyAsync(w->h_state, w->d_state, sizeof(CpdState), hipMemcpyDeviceToHost, c->stream));
            MI_HIP(hipEventRecord(c->peek_event, c->stream));
            MI_TRY(cpd_fgt_prelaunch(c, w, v));
            MI_HIP(hipEventSynchronize(c->peek_event));
        } else if (pipelined) {
            MI_HIP(hipMemcpyAsync(w->h_state, w->d_state, sizeof(CpdState), hipMemcpyDeviceToHost, c->stream));
            MI_HIP(hipEventRecord(c->peek_event, c->stream));
            MI_TRY(cpd_estep_enqueue(c, w, v));                    // (returns at once on the device if the state copied above says "done")
            MI_TRY(cpd_mstep_enqueue(c, w, v, rules, 1));
            MI_HIP(cpd_transform(v, w->m_pad, c->stream));
            ahead = true;
            MI_HIP(hipEventSynchronize(c->peek_event));
        } else
            MI_TRY(cpd_fetch(c, w));
        if (params->verbose) printf("loop_nr %d, error: %f\n", w->h_state->iterations, w->h_state->error);
    }
    // The host checks above only PEEK: behind the state copy that said "done" the stream still holds the iteration enqueued ahead of it (kernels
    // that return at once) or the prelaunched K-centre replay.  Drained here, so that the call's wall time brackets all of its device work
    // (bench.py's timers around this call used to miss that tail) and the workspace is quiet when the caller gets it back.
    MI_HIP(hipStreamSynchronize(c->stream));
    const CpdState* s = w->h_state;
    // return make_pair(scale * rotationMatrix, translationVector)   coherentpointdrift.cpp:123 / cpdcuda.cu:360
    for (int col = 0; col < 3; col++) {
        for (int row = 0; row < 3; row++) out_sR_t[4 * col + row] = s->scale * s->R[3 * col + row];
        out_sR_t[4 * col + 3] = 0.f;
    }
    out_sR_t[12] = s->t[0]; out_sR_t[13] = s->t[1]; out_sR_t[14] = s->t[2]; out_sR_t[15] = 1.f;
    if (out_scale) *out_scale = s->scale;
    *iterations = s->iterations;
    *error = s->error;
    return MI_OK;
}

extern "C" int mi_cpd_sigma_squared(mi_ctx* c, const float* before_xyz, int m, const float* after_xyz, int n, float* sigma2)
{
    return mi_cpd_sigma_squared_mode(c, before_xyz, m, after_xyz, n, MI_SIGMA2_EXACT, sigma2);
}

extern "C" int mi_cpd_sigma_squared_mode(mi_ctx* c, const float* before_xyz, int m, const float* after_xyz, int n, int sigma2_mode,
                                         float* sigma2)
{
    MI_TRY(cpd_check(c, before_xyz, m, after_xyz, n));
    if (!sigma2) { set_error("mi_cpd_sigma_squared: null output"); return MI_ERR_INVALID_ARG; }
    if (sigma2_mode != MI_SIGMA2_EXACT && sigma2_mode != MI_SIGMA2_CPU_SEQUENTIAL) { set_error("mi_cpd_sigma_squared: bad sigma2_mode %d", sigma2_mode); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    CpdWorkspace* w = nullptr;
    MI_TRY(cpd_workspace(c, &w));
    w->replicated = false;
    w->queries_sharded = false;
    MI_TRY(cpd_load(c, w, before_xyz, m, after_xyz, n));
    const CpdView v = cpd_view(c, w);
    mi_cpd_params p;
    mi_cpd_params_default(&p);
    p.max_iterations = 1;
    MI_TRY(cpd_init(c, w, v, cpd_rules(w, &p), 0.f, sigma2_mode));
    MI_TRY(cpd_fetch(c, w));
    *sigma2 = w->h_state->sigma2_init;
    return MI_OK;
}

// The three E-step flavours behind one primitive: mode 0 exact, 1 truncated exact, 2 Fast Gauss Transform.
static int estep_primitive(mi_ctx* c, const float* y_xyz, int m, const float* x_xyz, int n, int mode, float constant, float sigma2,
                           float truncate, float weight, float sigma2_init, float ratio_of_far_field, int order,
                           float* p1, float* pt1, float* px, float* L)
{
    MI_TRY(cpd_check(c, y_xyz, m, x_xyz, n));
    if (!p1 || !pt1 || !px || !L) { set_error("CPD E-step: null output"); return MI_ERR_INVALID_ARG; }
    if (!(sigma2 > 0.f)) { set_error("CPD E-step: sigma2 must be positive"); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    CpdWorkspace* w = nullptr;
    MI_TRY(cpd_workspace(c, &w));
    w->replicated = false;
    w->queries_sharded = false;
    MI_TRY(cpd_load(c, w, y_xyz, m, x_xyz, n));
    CpdView v = cpd_view(c, w);
    memset(w->h_state, 0, sizeof(CpdState));
    w->h_state->sigma2 = sigma2;
    w->h_state->constant = constant;
    w->h_state->scale = 1.f;
    MI_HIP(hipMemcpyAsync(w->d_state, w->h_state, sizeof(CpdState), hipMemcpyHostToDevice, c->stream));
    if (mode == 2) {
        MI_TRY(cpd_estep_fgt_enqueue(c, w, v, weight, sigma2, sigma2_init, ratio_of_far_field, order));
    } else {
        if (mode == 1) {
            if (!(truncate > 0.f)) { set_error("mi_cpd_estep_truncated: truncate must be positive"); return MI_ERR_INVALID_ARG; }
            v.truncate = 1;
            v.trunc_log = std::log(truncate);
        }
        if (mode == 1 && c->tune.cpd_trunc_cull != 0) MI_TRY(cpd_estep_trunc_enqueue(c, w, v));
        else MI_TRY(cpd_estep_enqueue(c, w, v));
    }
    w->sums_fresh = false;               // (a stand-alone E-step: nothing of it is carried into a later M-step call)
    const int nxb = cpd_standalone_sum_blocks(n);
    MI_HIP(cpd_xsums(v, w->part_x.p, nxb, c->stream));
    MI_HIP(hipMemcpyAsync(p1, w->p1.p, sizeof(float) * (size_t)m, hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipMemcpyAsync(pt1, w->pt1.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipMemcpyAsync(px, w->px.p, sizeof(float) * 3 * (size_t)m, hipMemcpyDeviceToHost, c->stream));
    std::vector<double> part((size_t)nxb * CPD_XSUMS);
    MI_HIP(hipMemcpyAsync(part.data(), w->part_x.p, sizeof(double) * part.size(), hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipStreamSynchronize(c->stream));
    double logsum = 0.0;
    for (int b = 0; b < nxb; b++) logsum += part[(size_t)b * CPD_XSUMS];
    *L = (float)(-logsum) + (float)(3 * n) * logf(sigma2) / 2.0f;     // coherentpointdrift.cpp:215-217 / cpdutils.cpp:69-72
    return MI_OK;
}

extern "C" int mi_cpd_estep(mi_ctx* c, const float* y_xyz, int m, const float* x_xyz, int n, float constant, float sigma2,
                            float* p1, float* pt1, float* px, float* L)
{
    return estep_primitive(c, y_xyz, m, x_xyz, n, 0, constant, sigma2, 0.f, 0.f, 0.f, 0.f, 0, p1, pt1, px, L);
}

extern "C" int mi_cpd_estep_truncated(mi_ctx* c, const float* y_xyz, int m, const float* x_xyz, int n, float constant, float sigma2,
                                      float truncate, float* p1, float* pt1, float* px, float* L)
{
    return estep_primitive(c, y_xyz, m, x_xyz, n, 1, constant, sigma2, truncate, 0.f, 0.f, 0.f, 0, p1, pt1, px, L);
}

extern "C" int mi_cpd_estep_fgt(mi_ctx* c, const float* y_xyz, int m, const float* x_xyz, int n, float weight, float sigma2,
                                float sigma2_init, float ratio_of_far_field, int order_of_truncation,
                                float* p1, float* pt1, float* px, float* L)
{
    if (!(sigma2_init > 0.f)) { set_error("mi_cpd_estep_fgt: sigma2_init must be positive"); return MI_ERR_INVALID_ARG; }
    if (!(weight > 0.f && weight < 1.f)) { set_error("mi_cpd_estep_fgt: weight must lie in (0, 1)"); return MI_ERR_INVALID_ARG; }
    return estep_primitive(c, y_xyz, m, x_xyz, n, 2, 0.f, sigma2, 0.f, weight, sigma2_init, ratio_of_far_field, order_of_truncation,
                           p1, pt1, px, L);
}

extern "C" int mi_fgt_kcenter(mi_ctx* c, const float* cloud_xyz, int n, int K, float* centers_xyz, int* cluster)
{
    if (!c) { set_error("mi_fgt_kcenter: null context"); return MI_ERR_INVALID_ARG; }
    if (!cloud_xyz || !centers_xyz || !cluster) { set_error("mi_fgt_kcenter: null argument"); return MI_ERR_INVALID_ARG; }
    if (n < 2 || K < 1 || K > FGT_MAX_CLUSTERS) { set_error("mi_fgt_kcenter: need n >= 2 and 1 <= K <= %d (n=%d, K=%d)", FGT_MAX_CLUSTERS, n, K); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    CpdWorkspace* w = nullptr;
    MI_TRY(cpd_workspace(c, &w));
    c->icp_loaded = false;
    const int n_pad = round_up_i(n, NN_SRC_PAD);
    MI_TRY(c->bx.reserve(n_pad)); MI_TRY(c->by.reserve(n_pad)); MI_TRY(c->bz.reserve(n_pad));
    MI_TRY(upload_soa(c, cloud_xyz, n, n_pad, c->bx.p, c->by.p, c->bz.p, nullptr));
    FgtClusters cl{};
    w->fgt.y.guess_K = 0;                // (the buffers of the moving side now hold another cloud's sweep)
    MI_TRY(fgt_side(c, &w->fgt, &w->fgt.y, c->bx.p, c->by.p, c->bz.p, n, K, &cl));
    MI_HIP(fgt_cluster(cl, w->fgt.sort_temp.p, w->fgt.sort_temp.cap, c->stream));
    MI_HIP(hipMemcpyAsync(cluster, cl.indx, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipMemcpyAsync(centers_xyz, cl.xc, sizeof(float) * 3 * (size_t)K, hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipStreamSynchronize(c->stream));
    return MI_OK;
}

extern "C" int mi_fgt_kcenter_guided(mi_ctx* c, const float* cloud_xyz, int n, int K, const int* guess, int n_guess, float* centers_xyz, int* cluster,
                                     int* picked, int* verified)
{
    if (!c) { set_error("mi_fgt_kcenter_guided: null context"); return MI_ERR_INVALID_ARG; }
    if (!cloud_xyz || !centers_xyz || !cluster || (n_guess > 0 && !guess)) { set_error("mi_fgt_kcenter_guided: null argument"); return MI_ERR_INVALID_ARG; }
    if (n < 2 || K < 1 || K > FGT_MAX_CLUSTERS || n_guess < 0) { set_error("mi_fgt_kcenter_guided: need n >= 2, 1 <= K <= %d, n_guess >= 0 (n=%d, K=%d, n_guess=%d)", FGT_MAX_CLUSTERS, n, K, n_guess); return MI_ERR_INVALID_ARG; }
    for (int i = 0; i < n_guess; i++)
        if (guess[i] < 0 || guess[i] >= n) { set_error("mi_fgt_kcenter_guided: guess[%d] = %d is not a point of the cloud", i, guess[i]); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    CpdWorkspace* w = nullptr;
    MI_TRY(cpd_workspace(c, &w));
    c->icp_loaded = false;
    w->fgt.y.guess_K = 0;
    const int n_pad = round_up_i(n, NN_SRC_PAD);
    MI_TRY(c->bx.reserve(n_pad)); MI_TRY(c->by.reserve(n_pad)); MI_TRY(c->bz.reserve(n_pad));
    MI_TRY(upload_soa(c, cloud_xyz, n, n_pad, c->bx.p, c->by.p, c->bz.p, nullptr));
    FgtClusters cl{};
    MI_TRY(fgt_side(c, &w->fgt, &w->fgt.y, c->bx.p, c->by.p, c->bz.p, n, K, &cl));
    const int g = std::min(n_guess, K);
    if (g > 0) MI_HIP(hipMemcpyAsync(cl.picked, guess, sizeof(int) * (size_t)g, hipMemcpyHostToDevice, c->stream));
    MI_HIP(hipMemsetAsync(cl.replay_state, 0xff, sizeof(int), c->stream));            // (-1: no replay ran)
    MI_TRY(fgt_arm_replay(c, &w->fgt.y, &cl, g));
    MI_HIP(fgt_cluster(cl, w->fgt.sort_temp.p, w->fgt.sort_temp.cap, c->stream));
    MI_HIP(hipMemcpyAsync(cluster, cl.indx, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipMemcpyAsync(centers_xyz, cl.xc, sizeof(float) * 3 * (size_t)K, hipMemcpyDeviceToHost, c->stream));
    if (picked) MI_HIP(hipMemcpyAsync(picked, cl.picked, sizeof(int) * (size_t)K, hipMemcpyDeviceToHost, c->stream));
    int state = -1;
    MI_HIP(hipMemcpyAsync(&state, cl.replay_state, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipStreamSynchronize(c->stream));
    if (verified) *verified = state;
    return MI_OK;
}

// Host-only: the monomial tables the FGT kernels use (exponents a | b<<8 | c<<16, C_k, Horner slot), for the CPU-side tests.
extern "C" int mi_fgt_tables(int order_of_truncation, unsigned int* mono, float* ck, int* horner_slot, int* pd_out)
{
    if (order_of_truncation < 1 || order_of_truncation > FGT_MAX_ORDER) { set_error("mi_fgt_tables: order outside [1, %d]", FGT_MAX_ORDER); return MI_ERR_INVALID_ARG; }
    std::vector<unsigned int> m; std::vector<float> k; std::vector<int> h;
    fgt_build_tables(order_of_truncation, m, k, h);
    if (pd_out) *pd_out = (int)m.size();
    if (mono) memcpy(mono, m.data(), m.size() * sizeof(unsigned int));
    if (ck) memcpy(ck, k.data(), k.size() * sizeof(float));
    if (horner_slot) memcpy(horner_slot, h.data(), h.size() * sizeof(int));
    return MI_OK;
}

extern "C" int mi_cpd_mstep(mi_ctx* c, const float* before_xyz, int m, const float* after_xyz, int n, const float* p1,
                            const float* pt1, const float* px, int const_scale, float out_R9[9], float out_t3[3], float* scale,
                            float* sigma2)
{
    MI_TRY(cpd_check(c, before_xyz, m, after_xyz, n));
    if (!p1 || !pt1 || !px || !out_R9 || !out_t3 || !scale || !sigma2) { set_error("mi_cpd_mstep: null argument"); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    CpdWorkspace* w = nullptr;
    MI_TRY(cpd_workspace(c, &w));
    w->replicated = false;
    w->queries_sharded = false;
    MI_TRY(cpd_load(c, w, before_xyz, m, after_xyz, n));
    CpdView v = cpd_view(c, w);
    v.xw4 = nullptr;   // no E-step ran: skip the log-likelihood term
    w->sums_fresh = false;
    MI_HIP(hipMemcpyAsync(w->p1.p, p1, sizeof(float) * (size_t)m, hipMemcpyHostToDevice, c->stream));
    MI_HIP(hipMemcpyAsync(w->pt1.p, pt1, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    MI_HIP(hipMemcpyAsync(w->px.p, px, sizeof(float) * 3 * (size_t)m, hipMemcpyHostToDevice, c->stream));
    memset(w->h_state, 0, sizeof(CpdState));
    w->h_state->scale = *scale;
    w->h_state->sigma2 = *sigma2;
    MI_HIP(hipMemcpyAsync(w->d_state, w->h_state, sizeof(CpdState), hipMemcpyHostToDevice, c->stream));
    mi_cpd_params p;
    mi_cpd_params_default(&p);
    p.const_scale = const_scale;
    MI_TRY(cpd_mstep_enqueue(c, w, v, cpd_rules(w, &p), 0));
    MI_TRY(cpd_fetch(c, w));
    memcpy(out_R9, w->h_state->R, sizeof(float) * 9);
    memcpy(out_t3, w->h_state->t, sizeof(float) * 3);
    *scale = w->h_state->scale;
    *sigma2 = w->h_state->sigma2;
    return MI_OK;
}
