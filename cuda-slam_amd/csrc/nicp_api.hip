// C ABI, non-iterative registration ("method": "nicp"): replaces GetCudaNicpTransformationMatrix (source/cuda-slam/nicpcuda.cu:70-184,
// parallelsvdhelper.cu:60-79); oracle NonIterative::GetNonIterativeTransformationMatrix, sequential policy
// (source/cpu-slam/noniterative.cpp:204-282) + GetSingleNonIterativeSlamResult (:25-55).
//
// The method aligns the principal axes of the two clouds: rotation = U_after * U_before^T with U from an SVD of the centred
// 3 x N matrices, repeated over random permutations of the clouds, keeping the candidate with the smallest error.  The
// reference runs one N x 3 gesvd per cloud and repetition (batched from std::threads on the GPU, Eigen::JacobiSVD on the CPU).
// Here the O(N) work is done ONCE:
//   * U of a 3 x N matrix does not depend on the column order; only the SIGNS of its columns do -- Eigen's SVD preconditions
//     with a column-pivoted Householder QR of the transpose, and each reflector's sign is minus the sign of the pivot entry in
//     the FIRST remaining row.  The R factor (3 x 3) of that QR is a closed-form function of the Gram matrix A A^T and of the
//     first three columns of A (the first three points of the permuted cloud) alone: reflecting column j against reflector k
//     only mixes column k into it, so the trailing Gram matrix and the leading rows can be carried along exactly.
//   * so one fused moments kernel (centroids, both Gram matrices, the pair sums the "approximated error" needs; fp64, fixed
//     order) feeds an O(1) solve per repetition: QR from the Gram matrix, the same two-sided Jacobi as K3 on R^T, sign fix, sort.
//   * the candidates' errors on the comparison subcloud reuse K1 / K1t (transform, exact nearest neighbours, mean d^2).
// The permutations themselves are the caller's (the host side draws them from mt19937 + std::shuffle exactly as the reference
// does); the library only needs the first three indices of each and the subcloud's indices.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "context.h"
#include "kernels.h"
#include "reduce.hpp"
#include "svd3.hpp"

using namespace mislam;

namespace mislam {

constexpr int NICP_SUMS = 32;
// 0-2 sum b | 3-8 sum b b^T (xx xy xz yy yz zz) | 9-11 sum a | 12-17 sum a a^T | pairs i < min(m, n): 18-20 sum a_i, 21 sum |a_i|^2,
// 22-30 sum a_i b_i^T (row-major in a) | 31 spare

__global__ __launch_bounds__(256) void nicp_moments_kernel(const float* __restrict__ bx, const float* __restrict__ by, const float* __restrict__ bz,
                                                           int m, const float* __restrict__ ax, const float* __restrict__ ay,
                                                           const float* __restrict__ az, int n, double* __restrict__ partials)
{
    double acc[NICP_SUMS] = {0};
    const int top = max(m, n);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < top; i += gridDim.x * 256) {
        double b[3] = {0, 0, 0}, a[3] = {0, 0, 0};
        if (i < m) {
            b[0] = bx[i]; b[1] = by[i]; b[2] = bz[i];
            acc[0] += b[0]; acc[1] += b[1]; acc[2] += b[2];
            acc[3] += b[0] * b[0]; acc[4] += b[0] * b[1]; acc[5] += b[0] * b[2]; acc[6] += b[1] * b[1]; acc[7] += b[1] * b[2]; acc[8] += b[2] * b[2];
        }
        if (i < n) {
            a[0] = ax[i]; a[1] = ay[i]; a[2] = az[i];
            acc[9] += a[0]; acc[10] += a[1]; acc[11] += a[2];
            acc[12] += a[0] * a[0]; acc[13] += a[0] * a[1]; acc[14] += a[0] * a[2]; acc[15] += a[1] * a[1]; acc[16] += a[1] * a[2]; acc[17] += a[2] * a[2];
        }
        if (i < m && i < n) {
            acc[18] += a[0]; acc[19] += a[1]; acc[20] += a[2];
            acc[21] += a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) acc[22 + 3 * r + c] += a[r] * b[c];
        }
    }
    block_sum_store<NICP_SUMS>(acc, partials + (size_t)blockIdx.x * NICP_SUMS);
}

__global__ __launch_bounds__(256) void nicp_reduce_kernel(const double* __restrict__ partials, int nblocks, double* __restrict__ out)
{
    __shared__ double lds[256];
    double s[NICP_SUMS];
    reduce_partials<NICP_SUMS>(partials, nblocks, s, lds);
    if (threadIdx.x < NICP_SUMS) out[threadIdx.x] = s[threadIdx.x];
}

struct Rt {
    float R[9];   // column-major (glm::mat3)
    float t[3];
};

// TransformPoint(point, R, t) = (R * p) + t in glm's operation order (common.cpp:45-49)
__global__ __launch_bounds__(256) void nicp_transform_kernel(const float* __restrict__ bx, const float* __restrict__ by,
                                                             const float* __restrict__ bz, int n_pad, Rt rt, float* __restrict__ cx,
                                                             float* __restrict__ cy, float* __restrict__ cz)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pad) return;
    const float x = bx[i], y = by[i], z = bz[i];
    cx[i] = ((rt.R[0] * x + rt.R[3] * y) + rt.R[6] * z) + rt.t[0];
    cy[i] = ((rt.R[1] * x + rt.R[4] * y) + rt.R[7] * z) + rt.t[1];
    cz[i] = ((rt.R[2] * x + rt.R[5] * y) + rt.R[8] * z) + rt.t[2];
}

// sum of the matched squared distances below max_d2 and their count (GetCorrespondingPoints + GetMeanSquaredError, noniterative.cpp:229-231)
__global__ __launch_bounds__(256) void nicp_error_kernel(const unsigned long long* __restrict__ keys, int n, float max_d2, double* __restrict__ partials)
{
    double acc[2] = {0, 0};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float d2 = __uint_as_float((unsigned int)(keys[i] >> 32));
        if (d2 < max_d2) { acc[0] += (double)d2; acc[1] += 1.0; }
    }
    block_sum_store<2>(acc, partials + (size_t)blockIdx.x * 2);
}

// ---------------------------------------------------------------------------------------------------------------
// host-side O(1) solve
// ---------------------------------------------------------------------------------------------------------------
// R factor and column permutation of Eigen's ColPivHouseholderQR (ColPivHouseholderQR.h:478-580, Householder.h:65-97) of the
// N x 3 matrix A whose Gram matrix A^T A is G and whose first three rows are `rows` (see the header of this file).
static void qr_from_gram(const double G_in[3][3], const double rows_in[3][3], double Rm[3][3], double P[3][3])
{
    double G[3][3], r[3][3];
    memcpy(G, G_in, sizeof G);
    memcpy(r, rows_in, sizeof r);
    memset(Rm, 0, sizeof(double) * 9);
    double upd[3], direct[3];
    int trans[3];
    for (int j = 0; j < 3; j++) upd[j] = direct[j] = std::sqrt(std::max(G[j][j], 0.0));
    const double downdate_threshold = std::sqrt((double)FLT_EPSILON);
    auto swap_cols = [&](int p, int q) {
        for (int i = 0; i < 3; i++) { std::swap(G[i][p], G[i][q]); }
        for (int j = 0; j < 3; j++) { std::swap(G[p][j], G[q][j]); }
        for (int i = 0; i < 3; i++) { std::swap(r[i][p], r[i][q]); std::swap(Rm[i][p], Rm[i][q]); }
        std::swap(upd[p], upd[q]); std::swap(direct[p], direct[q]);
    };
    for (int k = 0; k < 3; k++) {
        int big = k;
        for (int j = k + 1; j < 3; j++) if (upd[j] > upd[big]) big = j;
        trans[k] = big;
        if (big != k) swap_cols(k, big);
        // G is the Gram matrix of rows >= k of the current (reflected, swapped) matrix; r[k] is its row k
        const double c0 = r[k][k];
        const double tail_sq = G[k][k] - c0 * c0;
        double beta, tau, inv;
        if (tail_sq <= (double)FLT_MIN) { tau = 0.0; beta = c0; inv = 0.0; }
        else {
            beta = std::sqrt(c0 * c0 + tail_sq);
            if (c0 >= 0.0) beta = -beta;
            inv = 1.0 / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        Rm[k][k] = beta;
        double alpha[3] = {0, 0, 0};
        for (int j = k + 1; j < 3; j++) {
            const double w = r[k][j] + (G[k][j] - c0 * r[k][j]) * inv;    // v^T col_j with v = (1, tail / (c0 - beta))
            Rm[k][j] = r[k][j] - tau * w;
            alpha[j] = tau * w * inv;                                      // rows > k: col_j <- col_j - alpha_j col_k
        }
        double G1[3][3], Gn[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) G1[i][j] = G[i][j] - r[k][i] * r[k][j];   // drop row k
        memcpy(Gn, G1, sizeof Gn);
        for (int j = k + 1; j < 3; j++)
            for (int l = k + 1; l < 3; l++)
                Gn[j][l] = G1[j][l] - alpha[l] * G1[j][k] - alpha[j] * G1[k][l] + alpha[j] * alpha[l] * G1[k][k];
        for (int q = k + 1; q < 3; q++) {
            const double rk = r[q][k];
            for (int j = k + 1; j < 3; j++) r[q][j] -= alpha[j] * rk;
        }
        memcpy(G, Gn, sizeof G);
        for (int j = k + 1; j < 3; j++) {                                  // the norm downdate that steers the next pivot, :551-570
            if (upd[j] != 0.0) {
                double temp = std::fabs(Rm[k][j]) / upd[j];
                temp = (1.0 + temp) * (1.0 - temp);
                if (temp < 0.0) temp = 0.0;
                const double ratio = upd[j] / direct[j];
                if (temp * ratio * ratio <= downdate_threshold) { direct[j] = std::sqrt(std::max(G[j][j], 0.0)); upd[j] = direct[j]; }
                else upd[j] *= std::sqrt(temp);
            }
        }
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) P[i][j] = i == j ? 1.0 : 0.0;
    for (int k = 0; k < 3; k++)
        if (trans[k] != k)
            for (int i = 0; i < 3; i++) std::swap(P[i][k], P[i][trans[k]]);
}

// matrixU() of JacobiSVD<Matrix3Xf>(centred cloud as columns, ComputeThinU | ComputeThinV): JacobiSVD.h:360-395, :683-781
static void svd_u_from_gram(const double G[3][3], const double rows[3][3], double U[3][3])
{
    // any positive scale leaves the signs alone; Eigen divides by the largest |entry|, the largest column norm serves as well
    double s = 0.0;
    for (int j = 0; j < 3; j++) s = std::max(s, std::sqrt(std::max(G[j][j], 0.0)));
    if (s == 0.0) s = 1.0;
    double Gs[3][3], rs[3][3], Rm[3][3], P[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Gs[i][j] = G[i][j] / (s * s); rs[i][j] = rows[i][j] / s; }
    qr_from_gram(Gs, rs, Rm, P);
    Mat3 W;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) W.a[i][j] = (float)Rm[j][i];   // work matrix = R^adjoint
    const Svd3 sv = svd3(W);                                                              // same Jacobi scheme, sign fix and sort as K3
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double acc = 0.0;
            for (int k = 0; k < 3; k++) acc += P[i][k] * (double)sv.U.a[k][j];             // U starts as the column permutation
            U[i][j] = acc;
        }
}

struct NicpSums {
    double cb[3], ca[3];          // centroids
    double Gb[3][3], Ga[3][3];    // Gram matrices of the centred clouds
    double Saa;                   // sum |a_i - ca|^2 over the pairs i < m
    double Sab[3][3];             // sum (a_i - ca)(b_i - cb)^T over the pairs
    int m, n;
};

static void sums_from_raw(const double* s, int m, int n, NicpSums* o)
{
    o->m = m; o->n = n;
    for (int d = 0; d < 3; d++) { o->cb[d] = s[d] / m; o->ca[d] = s[9 + d] / n; }
    const int sym[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            o->Gb[i][j] = s[3 + sym[i][j]] - m * o->cb[i] * o->cb[j];
            o->Ga[i][j] = s[12 + sym[i][j]] - n * o->ca[i] * o->ca[j];
        }
    const int pairs = std::min(m, n);
    o->Saa = s[21] - 2.0 * (o->ca[0] * s[18] + o->ca[1] * s[19] + o->ca[2] * s[20])
           + pairs * (o->ca[0] * o->ca[0] + o->ca[1] * o->ca[1] + o->ca[2] * o->ca[2]);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            o->Sab[r][c] = s[22 + 3 * r + c] - o->ca[r] * s[c] - s[18 + r] * o->cb[c] + pairs * o->ca[r] * o->cb[c];
}

struct NicpCandidate {
    Rt rt;
    float approx;
};

// GetSingleNonIterativeSlamResult (noniterative.cpp:25-55) for the permutation whose first three indices are head[0..2]
static NicpCandidate nicp_candidate(const NicpSums& s, const float* before, const float* after, const int head[3])
{
    double rb[3][3], ra[3][3], Ub[3][3], Ua[3][3], R[3][3];
    for (int i = 0; i < 3; i++)
        for (int d = 0; d < 3; d++) {
            rb[i][d] = (double)before[3 * (size_t)head[i] + d] - s.cb[d];
            ra[i][d] = (double)after[3 * (size_t)head[i] + d] - s.ca[d];
        }
    svd_u_from_gram(s.Gb, rb, Ub);
    svd_u_from_gram(s.Ga, ra, Ua);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double acc = 0.0;
            for (int k = 0; k < 3; k++) acc += Ua[i][k] * Ub[j][k];                        // U_after * U_before^T, :46
            R[i][j] = acc;
        }
    NicpCandidate c{};
    float Rf[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Rf[i][j] = (float)R[i][j]; c.rt.R[3 * j + i] = Rf[i][j]; }
    for (int i = 0; i < 3; i++)                                                           // t = centerAfter - R * centerBefore, :49
        c.rt.t[i] = (float)(s.ca[i] - (((double)Rf[i][0] * s.cb[0] + (double)Rf[i][1] * s.cb[1]) + (double)Rf[i][2] * s.cb[2]));
    // approximated error = mean_i |a'_i - R b'_i|^2 over index-wise pairs (common.cpp:233-244), expanded in the moments
    double rtr_gb = 0.0, cross = 0.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double rtr = 0.0;
            for (int k = 0; k < 3; k++) rtr += (double)Rf[k][i] * (double)Rf[k][j];
            rtr_gb += rtr * s.Gb[i][j];
            cross += (double)Rf[i][j] * s.Sab[i][j];
        }
    c.approx = (float)((s.Saa + rtr_gb - 2.0 * cross) / (double)s.m);
    return c;
}

// StoreResultIfOptimal with its quirks (common/nicputils.cpp:5-26): a result that beats several stored ones is inserted before
// each of them in turn; the list is cut back only when an insertion overflows it
static void store_if_optimal(std::vector<NicpCandidate>& list, const NicpCandidate& r, int desired)
{
    const int len0 = (int)list.size();
    if (len0 == 0 && desired > 0) { list.push_back(r); return; }
    for (int i = 0; i < len0; i++) {
        if (r.approx < list[i].approx) {
            list.insert(list.begin() + i, r);
            if ((int)list.size() > desired) { list.resize(desired); return; }
        }
    }
}

}  // namespace mislam

extern "C" void mi_nicp_params_default(mi_nicp_params* p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->eps = 1e-3f;                 // "convergence-epsilon"  configparser.cpp:244
    p->max_repetitions = 32;        // "nicp-iterations"      configparser.cpp:234
    p->approximation = MI_CPD_APPROX_HYBRID;   // "approximation-type": the parser's default
    p->verbose = 0;
}

extern "C" int mi_nicp_register(mi_ctx* c, const float* before_xyz, int m_before, const float* after_xyz, int n_after,
                                const mi_nicp_params* params, const int* order_heads, const int* subcloud_idx, int subcloud_n,
                                float out_T[16], int* repetitions, float* error)
{
    if (!c) { set_error("mi_nicp_register: null context"); return MI_ERR_INVALID_ARG; }
    if (!before_xyz || !after_xyz || !params || !order_heads || !out_T || !repetitions || !error) { set_error("mi_nicp_register: null argument"); return MI_ERR_INVALID_ARG; }
    if (m_before < 3 || n_after < 3) { set_error("mi_nicp_register: clouds need at least 3 points (m=%d, n=%d)", m_before, n_after); return MI_ERR_INVALID_ARG; }
    if (c->world != 1) { set_error("mi_nicp_register: single-GPU contexts only"); return MI_ERR_STATE; }
    if (params->approximation < MI_CPD_APPROX_NONE || params->approximation > MI_CPD_APPROX_HYBRID) { set_error("mi_nicp_register: unknown approximation %d", params->approximation); return MI_ERR_INVALID_ARG; }
    if (params->approximation != MI_CPD_APPROX_NONE && m_before > n_after) {
        set_error("mi_nicp_register: the approximated error pairs point i with point i: needs |before| <= |after| (%d > %d)", m_before, n_after);
        return MI_ERR_INVALID_ARG;
    }
    int max_rep = params->max_repetitions;
    if (max_rep == -1) max_rep = 20;                                  // noniterative.cpp:207-208
    if (max_rep < 0) { set_error("mi_nicp_register: max_repetitions %d", max_rep); return MI_ERR_INVALID_ARG; }
    const int size = std::min(m_before, n_after);
    for (int i = 0; i < 3 * max_rep; i++)
        if (order_heads[i] < 0 || order_heads[i] >= size) { set_error("mi_nicp_register: order_heads[%d] = %d outside [0, %d)", i, order_heads[i], size); return MI_ERR_INVALID_ARG; }
    if (subcloud_n < 1 || subcloud_n > m_before) { set_error("mi_nicp_register: subcloud of %d points", subcloud_n); return MI_ERR_INVALID_ARG; }
    if (subcloud_idx)
        for (int i = 0; i < subcloud_n; i++)
            if (subcloud_idx[i] < 0 || subcloud_idx[i] >= m_before) { set_error("mi_nicp_register: subcloud_idx[%d] out of range", i); return MI_ERR_INVALID_ARG; }
    MI_ENTER(c);
    c->icp_loaded = false;

    // ---- one pass over both clouds: centroids, Gram matrices, pair sums
    const int m_pad = (m_before + NN_SRC_PAD - 1) / NN_SRC_PAD * NN_SRC_PAD;
    MI_TRY(c->bx.reserve(m_pad)); MI_TRY(c->by.reserve(m_pad)); MI_TRY(c->bz.reserve(m_pad));
    MI_TRY(upload_soa(c, before_xyz, m_before, m_pad, c->bx.p, c->by.p, c->bz.p, nullptr));
    MI_TRY(upload_target_shard(c, after_xyz, n_after));
    const int nb = icp_reduce_blocks(std::max(m_before, n_after));
    MI_TRY(c->part_mom.reserve((size_t)ICP_MAX_PARTIAL_BLOCKS * NICP_SUMS + NICP_SUMS));
    double* d_sums = c->part_mom.p + (size_t)ICP_MAX_PARTIAL_BLOCKS * NICP_SUMS;
    {
        ProfScope ps(c, MI_KERNEL_MOMENTS);
        hipLaunchKernelGGL(nicp_moments_kernel, dim3(nb), dim3(256), 0, c->stream, c->bx.p, c->by.p, c->bz.p, m_before, c->tx.p, c->ty.p,
                           c->tz.p, n_after, c->part_mom.p);
        hipLaunchKernelGGL(nicp_reduce_kernel, dim3(1), dim3(256), 0, c->stream, c->part_mom.p, nb, d_sums);
        MI_HIP(hipGetLastError());
    }
    double raw[NICP_SUMS];
    MI_HIP(hipMemcpyAsync(raw, d_sums, sizeof raw, hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipStreamSynchronize(c->stream));
    NicpSums sums;
    sums_from_raw(raw, m_before, n_after, &sums);

    // ---- the comparison subcloud becomes the moving cloud of the error evaluations
    const int sn = subcloud_n, sn_pad = (sn + NN_SRC_PAD - 1) / NN_SRC_PAD * NN_SRC_PAD;
    std::vector<float> sub(3 * (size_t)sn);
    for (int i = 0; i < sn; i++) memcpy(&sub[3 * (size_t)i], before_xyz + 3 * (size_t)(subcloud_idx ? subcloud_idx[i] : i), 3 * sizeof(float));
    MI_TRY(c->bx.reserve(sn_pad)); MI_TRY(c->by.reserve(sn_pad)); MI_TRY(c->bz.reserve(sn_pad));
    MI_TRY(c->cx.reserve(sn_pad)); MI_TRY(c->cy.reserve(sn_pad)); MI_TRY(c->cz.reserve(sn_pad));
    MI_TRY(c->keys.reserve(sn_pad));
    MI_TRY(c->part_err.reserve((size_t)ICP_MAX_PARTIAL_BLOCKS * 2));
    MI_TRY(upload_soa(c, sub.data(), sn, sn_pad, c->bx.p, c->by.p, c->bz.p, nullptr));
    const int eb = icp_reduce_blocks(sn);
    std::vector<double> epart((size_t)eb * 2);
    // A thousand cold queries are too few to hide the box hierarchy's dependent loads (1.5 ms against 10^6 fixed points); the
    // every-pair kernel does the same 10^9 pairs in 0.2 ms.  Same neighbours either way.
    const int nn_mode = (double)sn * (double)n_after < 4e9 ? MI_NN_BRUTEFORCE : MI_NN_AUTO;
    // error of one candidate on the subcloud: transform, exact nearest neighbours in `after`, mean squared distance (:226-230)
    auto exact_error = [&](const Rt& rt, float* out) -> int {
        hipLaunchKernelGGL(nicp_transform_kernel, dim3((sn_pad + 255) / 256), dim3(256), 0, c->stream, c->bx.p, c->by.p, c->bz.p, sn_pad, rt,
                           c->cx.p, c->cy.p, c->cz.p);
        MI_HIP(fill_keys(c->keys.p, sn, c->stream));
        MI_TRY(launch_nn(c, c->cx.p, c->cy.p, c->cz.p, sn, n_after, 0, 0, nullptr, nn_mode));
        hipLaunchKernelGGL(nicp_error_kernel, dim3(eb), dim3(256), 0, c->stream, c->keys.p, sn, 1e6f, c->part_err.p);   // maxDistanceForComparison, :216
        MI_HIP(hipGetLastError());
        MI_HIP(hipMemcpyAsync(epart.data(), c->part_err.p, sizeof(double) * epart.size(), hipMemcpyDeviceToHost, c->stream));
        MI_HIP(hipStreamSynchronize(c->stream));
        double sum = 0.0, kept = 0.0;
        for (int b = 0; b < eb; b++) { sum += epart[2 * (size_t)b]; kept += epart[2 * (size_t)b + 1]; }
        *out = (float)(sum / kept);
        return MI_OK;
    };

    // ---- GetNonIterativeTransformationMatrixSequential, noniterative.cpp:204-282
    NicpCandidate best{};
    std::vector<NicpCandidate> best_list;
    float min_error = FLT_MAX;
    bool early = false;
    *error = 0.f;
    *repetitions = max_rep;
    for (int rep = 0; rep < max_rep && !early; rep++) {
        const NicpCandidate cand = nicp_candidate(sums, before_xyz, after_xyz, order_heads + 3 * (size_t)rep);
        *error = cand.approx;
        if (params->approximation == MI_CPD_APPROX_NONE) {
            MI_TRY(exact_error(cand.rt, error));
            if (params->verbose) printf("repetition %d, error: %f\n", rep + 1, *error);
            if (*error < min_error) {
                min_error = *error;
                best = cand;
                if (min_error <= params->eps) { *repetitions = rep + 1; early = true; }     // :238-242
            }
        } else {
            store_if_optimal(best_list, cand, params->approximation == MI_CPD_APPROX_HYBRID ? 5 : 1);
        }
    }
    if (!early && params->approximation != MI_CPD_APPROX_NONE) {                            // :261-279
        min_error = FLT_MAX;
        for (size_t i = 0; i < best_list.size() && !early; i++) {
            MI_TRY(exact_error(best_list[i].rt, error));
            if (*error < min_error) {
                min_error = *error;
                best = best_list[i];
                if (min_error <= params->eps) early = true;
            }
        }
    }
    if (!early) *error = min_error;                                                         // :281
    for (int col = 0; col < 3; col++) {
        for (int row = 0; row < 3; row++) out_T[4 * col + row] = best.rt.R[3 * col + row];
        out_T[4 * col + 3] = 0.f;
    }
    out_T[12] = best.rt.t[0]; out_T[13] = best.rt.t[1]; out_T[14] = best.rt.t[2]; out_T[15] = 1.f;
    return MI_OK;
}

// Touching one kernel of this translation unit makes the runtime load its code object now (mi_ctx_create) instead of at the
// first launch inside a registration call (deferred loading: 5-16 ms per object, once).
namespace mislam {
__global__ void preload_nicp_api_kernel() {}
hipError_t preload_nicp_api()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(preload_nicp_api_kernel));
}
}  // namespace mislam
