// C ABI, CPD part: rigid CPD driver (replaces CudaCPD, source/cuda-slam/cpdcuda.cu:302-361) and the test-grade
// E-step / M-step / sigma^2 primitives.  The EM loop is enqueued on one stream; the host reads the state block back
// every `sync_every` iterations only to learn whether the device-side stop rule has fired.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "context.h"
#include "cpd_kernels.h"

using namespace mislam;

namespace mislam {

struct CpdWorkspace {
    DevBuf<float> ax, ay, az;            // fixed cloud, SoA
    DevBuf<float> den_part, pt1, p1_part, px_part, p1, px;
    DevBuf<float4> xw4;
    DevBuf<double> part_x, part_k, part_init;
    CpdState* d_state = nullptr;
    CpdState* h_state = nullptr;
    int m = 0, n = 0, m_pad = 0, n_pad = 0;
    int k_chunks = 1, k_chunk_len = 0, x_chunks = 1, x_chunk_len = 0;
    mi_cpd_params params{};
};

void cpd_workspace_destroy(mi_ctx* c)
{
    CpdWorkspace* w = c->cpd;
    if (!w) return;
    w->ax.release(); w->ay.release(); w->az.release();
    w->den_part.release(); w->pt1.release(); w->p1_part.release(); w->px_part.release(); w->p1.release(); w->px.release();
    w->xw4.release(); w->part_x.release(); w->part_k.release(); w->part_init.release();
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
    w->m = m; w->n = n;
    w->m_pad = round_up_i(m, NN_SRC_PAD);
    w->n_pad = round_up_i(n, NN_SRC_PAD);
    const size_t mp = (size_t)w->m_pad, np = (size_t)w->n_pad;
    MI_TRY(c->bx.reserve(mp)); MI_TRY(c->by.reserve(mp)); MI_TRY(c->bz.reserve(mp));
    MI_TRY(c->cx.reserve(mp)); MI_TRY(c->cy.reserve(mp)); MI_TRY(c->cz.reserve(mp));
    MI_TRY(w->ax.reserve(np)); MI_TRY(w->ay.reserve(np)); MI_TRY(w->az.reserve(np));
    plan_chunks(c, n, 2, m, &w->k_chunks, &w->k_chunk_len);
    plan_chunks(c, m, 2, n, &w->x_chunks, &w->x_chunk_len);
    MI_TRY(w->den_part.reserve((size_t)w->k_chunks * n));
    MI_TRY(w->pt1.reserve(np));
    MI_TRY(w->xw4.reserve(np));
    MI_TRY(w->p1_part.reserve((size_t)w->x_chunks * m));
    MI_TRY(w->px_part.reserve((size_t)w->x_chunks * 3 * m));
    MI_TRY(w->p1.reserve(mp));
    MI_TRY(w->px.reserve(3 * mp));
    MI_TRY(w->part_x.reserve((size_t)ICP_MAX_PARTIAL_BLOCKS * CPD_XSUMS));
    MI_TRY(w->part_k.reserve((size_t)ICP_MAX_PARTIAL_BLOCKS * CPD_KSUMS));
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
    r.m = w->m; r.n = w->n;
    return r;
}

static int use_mfma_contraction()
{
    const char* v = getenv("MISLAM_CPD_MFMA");
    return (v && *v) ? atoi(v) : 1;
}

static int cpd_estep_enqueue(mi_ctx* c, CpdWorkspace* w, const CpdView& v)
{
    { ProfScope ps(c, MI_KERNEL_CPD_DENOM); MI_HIP(cpd_denominators(v, c->stream)); }
    MI_HIP(cpd_post_denominators(v, c->stream));
    { ProfScope ps(c, MI_KERNEL_CPD_CONTRACT); MI_HIP(cpd_contract(v, use_mfma_contraction(), c->stream)); }
    MI_HIP(cpd_post_contract(v, c->stream));
    (void)w;
    return MI_OK;
}

static int cpd_mstep_enqueue(mi_ctx* c, CpdWorkspace* w, const CpdView& v, const CpdRules& rules, int update_loop_state)
{
    const int nxb = icp_reduce_blocks(w->n), nkb = icp_reduce_blocks(w->m);
    ProfScope ps(c, MI_KERNEL_CPD_MSTEP);
    MI_HIP(cpd_xsums(v, w->part_x.p, nxb, c->stream));
    MI_HIP(cpd_ksums(v, w->part_k.p, nkb, c->stream));
    MI_HIP(cpd_solve(w->d_state, w->part_x.p, nxb, w->part_k.p, nkb, rules, update_loop_state, c->stream));
    return MI_OK;
}

static int cpd_fetch(mi_ctx* c, CpdWorkspace* w)
{
    MI_HIP(hipMemcpyAsync(w->h_state, w->d_state, sizeof(CpdState), hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipStreamSynchronize(c->stream));
    return MI_OK;
}

static int cpd_init(mi_ctx* c, CpdWorkspace* w, const CpdView& v, const CpdRules& rules, float sigma2_override)
{
    const int nb = icp_reduce_blocks(std::max(w->m, w->n));
    MI_HIP(cpd_init_sums(v, w->part_init.p, nb, c->stream));
    MI_HIP(cpd_init_state(w->d_state, w->part_init.p, nb, rules, sigma2_override, c->stream));
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
}

static int cpd_check(mi_ctx* c, const float* b, int m, const float* a, int n)
{
    if (!c) { set_error("CPD: null context"); return MI_ERR_INVALID_ARG; }
    if (!b || !a || m <= 0 || n <= 0) { set_error("CPD: empty or null cloud (m=%d, n=%d)", m, n); return MI_ERR_INVALID_ARG; }
    if (c->world != 1) { set_error("CPD: multi-GPU contexts are not supported yet"); return MI_ERR_STATE; }
    return MI_OK;
}

extern "C" int mi_cpd_register(mi_ctx* c, const float* before_xyz, int m_before, const float* after_xyz, int n_after,
                               const mi_cpd_params* params, float out_sR_t[16], float* out_scale, int* iterations, float* error)
{
    MI_TRY(cpd_check(c, before_xyz, m_before, after_xyz, n_after));
    if (!params || !out_sR_t || !iterations || !error) { set_error("mi_cpd_register: null argument"); return MI_ERR_INVALID_ARG; }
    MI_HIP(hipSetDevice(c->device));
    CpdWorkspace* w = nullptr;
    MI_TRY(cpd_workspace(c, &w));
    w->params = *params;
    MI_TRY(cpd_load(c, w, before_xyz, m_before, after_xyz, n_after));
    const CpdView v = cpd_view(c, w);
    const CpdRules rules = cpd_rules(w, params);
    MI_TRY(cpd_init(c, w, v, rules, params->sigma2_init));
    MI_TRY(cpd_fetch(c, w));
    int batch = params->sync_every;
    if (batch <= 0) {
        const double pairs = (double)m_before * (double)n_after;
        batch = pairs >= 2e9 ? 1 : (pairs >= 2e8 ? 2 : 8);
    }
    while (!w->h_state->done) {
        for (int b = 0; b < batch; b++) {
            MI_TRY(cpd_estep_enqueue(c, w, v));
            MI_TRY(cpd_mstep_enqueue(c, w, v, rules, 1));
            MI_HIP(cpd_transform(v, w->m_pad, c->stream));
        }
        MI_TRY(cpd_fetch(c, w));
        if (params->verbose) printf("loop_nr %d, error: %f\n", w->h_state->iterations, w->h_state->error);
    }
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
    MI_TRY(cpd_check(c, before_xyz, m, after_xyz, n));
    if (!sigma2) { set_error("mi_cpd_sigma_squared: null output"); return MI_ERR_INVALID_ARG; }
    MI_HIP(hipSetDevice(c->device));
    CpdWorkspace* w = nullptr;
    MI_TRY(cpd_workspace(c, &w));
    MI_TRY(cpd_load(c, w, before_xyz, m, after_xyz, n));
    const CpdView v = cpd_view(c, w);
    mi_cpd_params p;
    mi_cpd_params_default(&p);
    p.max_iterations = 1;
    MI_TRY(cpd_init(c, w, v, cpd_rules(w, &p), 0.f));
    MI_TRY(cpd_fetch(c, w));
    *sigma2 = w->h_state->sigma2_init;
    return MI_OK;
}

extern "C" int mi_cpd_estep(mi_ctx* c, const float* y_xyz, int m, const float* x_xyz, int n, float constant, float sigma2,
                            float* p1, float* pt1, float* px, float* L)
{
    MI_TRY(cpd_check(c, y_xyz, m, x_xyz, n));
    if (!p1 || !pt1 || !px || !L) { set_error("mi_cpd_estep: null output"); return MI_ERR_INVALID_ARG; }
    if (!(sigma2 > 0.f)) { set_error("mi_cpd_estep: sigma2 must be positive"); return MI_ERR_INVALID_ARG; }
    MI_HIP(hipSetDevice(c->device));
    CpdWorkspace* w = nullptr;
    MI_TRY(cpd_workspace(c, &w));
    MI_TRY(cpd_load(c, w, y_xyz, m, x_xyz, n));
    const CpdView v = cpd_view(c, w);
    memset(w->h_state, 0, sizeof(CpdState));
    w->h_state->sigma2 = sigma2;
    w->h_state->constant = constant;
    w->h_state->scale = 1.f;
    MI_HIP(hipMemcpyAsync(w->d_state, w->h_state, sizeof(CpdState), hipMemcpyHostToDevice, c->stream));
    MI_TRY(cpd_estep_enqueue(c, w, v));
    const int nxb = icp_reduce_blocks(n);
    MI_HIP(cpd_xsums(v, w->part_x.p, nxb, c->stream));
    MI_TRY(c->part_mom.reserve((size_t)ICP_MAX_PARTIAL_BLOCKS * ICP_MOMENTS));
    MI_HIP(hipMemcpyAsync(p1, w->p1.p, sizeof(float) * (size_t)m, hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipMemcpyAsync(pt1, w->pt1.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipMemcpyAsync(px, w->px.p, sizeof(float) * 3 * (size_t)m, hipMemcpyDeviceToHost, c->stream));
    std::vector<double> part((size_t)nxb * CPD_XSUMS);
    MI_HIP(hipMemcpyAsync(part.data(), w->part_x.p, sizeof(double) * part.size(), hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipStreamSynchronize(c->stream));
    double logsum = 0.0;
    for (int b = 0; b < nxb; b++) logsum += part[(size_t)b * CPD_XSUMS];
    *L = (float)(-logsum) + (float)(3 * n) * logf(sigma2) / 2.0f;     // coherentpointdrift.cpp:215-217
    return MI_OK;
}

extern "C" int mi_cpd_mstep(mi_ctx* c, const float* before_xyz, int m, const float* after_xyz, int n, const float* p1,
                            const float* pt1, const float* px, int const_scale, float out_R9[9], float out_t3[3], float* scale,
                            float* sigma2)
{
    MI_TRY(cpd_check(c, before_xyz, m, after_xyz, n));
    if (!p1 || !pt1 || !px || !out_R9 || !out_t3 || !scale || !sigma2) { set_error("mi_cpd_mstep: null argument"); return MI_ERR_INVALID_ARG; }
    MI_HIP(hipSetDevice(c->device));
    CpdWorkspace* w = nullptr;
    MI_TRY(cpd_workspace(c, &w));
    MI_TRY(cpd_load(c, w, before_xyz, m, after_xyz, n));
    CpdView v = cpd_view(c, w);
    v.xw4 = nullptr;   // no E-step ran: skip the log-likelihood term
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
