/* TEST INFRASTRUCTURE ONLY -- never linked into, imported by or shipped with the product path.
 *
 * Plain-C restatement of the reference's non-iterative ("nicp") registration: source/cpu-slam/noniterative.cpp (the sequential
 * policy, :204-282 -- the parallel policy's threads race on the shared random generator and read a stale result slot, :84-101,
 * so only the sequential one is deterministic), source/common/nicputils.cpp, and the pieces of the vendored Eigen 3.3.7 its
 * SVD call goes through for a 3 x N matrix: include/Eigen/src/SVD/JacobiSVD.h (:683-704 scaling, :360-395 the column-pivoting
 * QR preconditioner for more columns than rows, :740-781 sign fix and sort), include/Eigen/src/QR/ColPivHouseholderQR.h
 * (:478-580) and include/Eigen/src/Householder/Householder.h (:65-97).
 *
 * What makes the method "random": U of a 3 x N matrix does not depend on the column order, but its SIGNS come out of the
 * Householder steps of the QR of the transpose, i.e. of the first three points of the permuted cloud.  Every repetition
 * therefore yields one of the 2^k sign combinations of the same principal-axis alignment, and the driver keeps the best.
 *
 * The random permutations are INPUTS (drawn by the tests from the reference's own generator through oracle/_ref, or by the
 * product's host side from std::mt19937 + std::shuffle like the reference): nothing here depends on a library's shuffle. */
#define _GNU_SOURCE 1
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "slam_oracle.h"

/* ColPivHouseholderQR of an N x 3 float matrix (row-major a[i*3 + j]) -> R (3 x 3 upper, row-major) and the column permutation
 * as a matrix P (3 x 3, row-major).  Column sums run sequentially in float (Eigen's are vectorised; the difference is far below
 * what decides a pivot or a sign). */
static void colpiv_householder_qr_nx3(float* a, int n, float R[9], float P[9])
{
    float upd[3], direct[3];
    int trans[3];
    for (int j = 0; j < 3; j++) {
        float s = 0.f;
        for (int i = 0; i < n; i++) s += a[3 * i + j] * a[3 * i + j];
        upd[j] = direct[j] = sqrtf(s);                                  /* ColPivHouseholderQR.h:497-502 */
    }
    const float downdate_threshold = sqrtf(FLT_EPSILON);                /* :505 */
    for (int k = 0; k < 3; k++) {
        int big = k;                                                    /* :515-518 */
        for (int j = k + 1; j < 3; j++) if (upd[j] > upd[big]) big = j;
        trans[k] = big;
        if (big != k) {                                                 /* :528-533 */
            for (int i = 0; i < n; i++) { const float t = a[3 * i + k]; a[3 * i + k] = a[3 * i + big]; a[3 * i + big] = t; }
            float t = upd[k]; upd[k] = upd[big]; upd[big] = t;
            t = direct[k]; direct[k] = direct[big]; direct[big] = t;
        }
        /* makeHouseholderInPlace on a(k.., k): Householder.h:65-97 */
        const float c0 = a[3 * k + k];
        float tail_sq = 0.f;
        for (int i = k + 1; i < n; i++) tail_sq += a[3 * i + k] * a[3 * i + k];
        float beta, tau;
        if (tail_sq <= FLT_MIN) {
            tau = 0.f; beta = c0;
            for (int i = k + 1; i < n; i++) a[3 * i + k] = 0.f;
        } else {
            beta = sqrtf(c0 * c0 + tail_sq);
            if (c0 >= 0.f) beta = -beta;
            const float d = c0 - beta;
            for (int i = k + 1; i < n; i++) a[3 * i + k] = a[3 * i + k] / d;
            tau = (beta - c0) / beta;
        }
        a[3 * k + k] = beta;                                            /* :540 */
        /* applyHouseholderOnTheLeft to the trailing columns: Householder.h:116-133 */
        for (int j = k + 1; j < 3; j++) {
            float w = a[3 * k + j];
            for (int i = k + 1; i < n; i++) w += a[3 * i + k] * a[3 * i + j];
            a[3 * k + j] -= tau * w;
            for (int i = k + 1; i < n; i++) a[3 * i + j] -= tau * w * a[3 * i + k];
        }
        for (int j = k + 1; j < 3; j++) {                               /* norm downdate, :551-570 */
            if (upd[j] != 0.f) {
                float temp = fabsf(a[3 * k + j]) / upd[j];
                temp = (1.f + temp) * (1.f - temp);
                if (temp < 0.f) temp = 0.f;
                const float ratio = upd[j] / direct[j];
                const float temp2 = temp * (ratio * ratio);
                if (temp2 <= downdate_threshold) {
                    float s = 0.f;
                    for (int i = k + 1; i < n; i++) s += a[3 * i + j] * a[3 * i + j];
                    direct[j] = sqrtf(s);
                    upd[j] = direct[j];
                } else {
                    upd[j] *= sqrtf(temp);
                }
            }
        }
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = (j >= i) ? a[3 * i + j] : 0.f;
    for (int i = 0; i < 9; i++) P[i] = (i % 4 == 0) ? 1.f : 0.f;         /* :573-575: P = T_0 T_1 T_2 applied on the right */
    for (int k = 0; k < 3; k++)
        if (trans[k] != k)
            for (int i = 0; i < 3; i++) { const float t = P[3 * i + k]; P[3 * i + k] = P[3 * i + trans[k]]; P[3 * i + trans[k]] = t; }
}

/* matrixU() of Eigen::JacobiSVD<Matrix3Xf>(M, ComputeThinU | ComputeThinV) for the 3 x n matrix whose columns are the points
 * of `cloud` minus `center` (noniterative.cpp:34-44).  U row-major 3 x 3. */
static void jacobi_svd_u_3xn(const float* cloud, int n, const float center[3], float U[9])
{
    float* a = (float*)malloc(sizeof(float) * 3 * (size_t)n);          /* the adjoint: n x 3 */
    float scale = 0.f;                                                  /* JacobiSVD.h:690-692 */
    for (int i = 0; i < n; i++)
        for (int d = 0; d < 3; d++) {
            a[3 * i + d] = cloud[3 * i + d] - center[d];               /* GetAlignedCloud, common.cpp:327-333 */
            if (fabsf(a[3 * i + d]) > scale) scale = fabsf(a[3 * i + d]);
        }
    if (scale == 0.f) scale = 1.f;
    for (int i = 0; i < 3 * n; i++) a[i] /= scale;                      /* :698 */
    float R[9], P[9], W[9], u[9], s[3], v[9];
    colpiv_householder_qr_nx3(a, n, R, P);                              /* :366-367 */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) W[3 * i + j] = R[3 * j + i];   /* work matrix = R^adjoint, :368 */
    oracle_jacobi_svd3(W, u, s, v);                                     /* the sweeps, sign fix and sort start from U = I there ... */
    for (int i = 0; i < 3; i++)                                         /* ... and from U = colsPermutation here (:375): U = P * u */
        for (int j = 0; j < 3; j++) {
            float acc = 0.f;
            for (int k = 0; k < 3; k++) acc += P[3 * i + k] * u[3 * k + j];
            U[3 * i + j] = acc;
        }
    free(a);
}

/* GetSingleNonIterativeSlamResult: noniterative.cpp:25-55.  rot9 column-major (glm). */
void oracle_nicp_single(const float* before, int m, const float* after, int n, float rot9[9], float trans3[3], float* approximated_error)
{
    float cb[3], ca[3], ub[9], ua[9], R[9];
    oracle_center_of_mass(before, m, cb);                               /* :31-32 */
    oracle_center_of_mass(after, n, ca);
    jacobi_svd_u_3xn(before, m, cb, ub);
    jacobi_svd_u_3xn(after, n, ca, ua);
    for (int i = 0; i < 3; i++)                                         /* rotation = U_after * U_before^T, :46 */
        for (int j = 0; j < 3; j++) {
            float acc = 0.f;
            for (int k = 0; k < 3; k++) acc += ua[3 * i + k] * ub[3 * j + k];
            R[3 * i + j] = acc;
        }
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) rot9[3 * c + r] = R[3 * r + c];   /* ConvertRotationMatrix, common.cpp:335-346 */
    for (int i = 0; i < 3; i++)                                         /* t = centerAfter - R * centerBefore, :49 (glm mat3 * vec3) */
        trans3[i] = ca[i] - ((R[3 * i] * cb[0] + R[3 * i + 1] * cb[1]) + R[3 * i + 2] * cb[2]);
    /* GetMeanSquaredError(alignedBefore, alignedAfter, mat4(R)): common.cpp:233-244, index-wise pairs, sequential float sum */
    float sum = 0.f;
    for (int i = 0; i < m; i++) {
        const float bx = before[3 * i] - cb[0], by = before[3 * i + 1] - cb[1], bz = before[3 * i + 2] - cb[2];
        float d2 = 0.f;
        float diff[3];
        for (int r = 0; r < 3; r++) {
            const float tr = (R[3 * r] * bx + R[3 * r + 1] * by) + (R[3 * r + 2] * bz + 0.f * 1.0f);   /* glm mat4 * vec4 */
            diff[r] = (after[3 * i + r] - ca[r]) - tr;
        }
        d2 = diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2];
        sum += d2;
    }
    *approximated_error = sum / (float)(size_t)m;
}

typedef struct {
    float R[9];     /* column-major */
    float t[3];
    float approx;
} nicp_result;

/* StoreResultIfOptimal: common/nicputils.cpp:5-26 -- including its quirks: a result that beats several stored ones is inserted
 * before each of them in turn, and the list only shrinks when an insertion overflows it. */
static void store_if_optimal(nicp_result* list, int* length, const nicp_result* r, int desired)
{
    const int len0 = *length;
    if (len0 == 0 && desired > 0) { list[0] = *r; *length = 1; return; }
    for (int i = 0; i < len0; i++) {
        if (r->approx < list[i].approx) {
            memmove(list + i + 1, list + i, sizeof(nicp_result) * (size_t)(*length - i));
            list[i] = *r;
            (*length)++;
            if (*length > desired) { *length = desired; return; }
        }
    }
}

/* Error of one candidate on the comparison subcloud: noniterative.cpp:226-230 / :263-267 */
static float subcloud_error(const float* subcloud, int sn, const nicp_result* r, const float* after, int n)
{
    float* tr = (float*)malloc(sizeof(float) * 3 * (size_t)sn);
    int* idx = (int*)malloc(sizeof(int) * (size_t)sn);
    float* d2 = (float*)malloc(sizeof(float) * (size_t)sn);
    oracle_transform_cloud(subcloud, sn, r->R, r->t, 1.0f, 0, tr);
    oracle_nn_search(tr, sn, after, n, 0, 0, idx, d2);
    float sum = 0.f;
    int kept = 0;
    for (int i = 0; i < sn; i++)
        if (d2[i] < 1e6f) {                                              /* maxDistanceForComparison, :216 */
            const float dx = after[3 * idx[i]] - tr[3 * i], dy = after[3 * idx[i] + 1] - tr[3 * i + 1], dz = after[3 * idx[i] + 2] - tr[3 * i + 2];
            sum += dx * dx + dy * dy + dz * dz;                          /* GetMeanSquaredError(pairs), common.cpp:270-279 */
            kept++;
        }
    free(tr); free(idx); free(d2);
    return sum / (float)(size_t)kept;
}

/* GetNonIterativeTransformationMatrixSequential: noniterative.cpp:204-282.
 * perms: max_repetitions permutations of 0..min(m,n)-1, one per repetition (GetRandomPermutationVector, :222);
 * subcloud_idx: the first subcloud_n entries of the permutation GetSubcloud drew before them (common.cpp:25-37), or NULL with
 * subcloud_n = m when the subcloud is the whole cloud (no permutation is drawn then).  approximation: 0 none, 1 full, 2 hybrid. */
void oracle_nicp(const float* before, int m, const float* after, int n, float eps, int max_repetitions, int approximation,
                 const int* subcloud_idx, int subcloud_n, const int* perms, float rot9[9], float trans3[3], int* repetitions, float* error)
{
    const int size = m < n ? m : n;
    if (max_repetitions == -1) max_repetitions = 20;                     /* :207-208 */
    float* sub = (float*)malloc(sizeof(float) * 3 * (size_t)subcloud_n);
    for (int i = 0; i < subcloud_n; i++) {
        const int s = subcloud_idx ? subcloud_idx[i] : i;
        memcpy(sub + 3 * i, before + 3 * s, sizeof(float) * 3);
    }
    float* pb = (float*)malloc(sizeof(float) * 3 * (size_t)m);
    float* pa = (float*)malloc(sizeof(float) * 3 * (size_t)n);
    nicp_result best_list[8];
    int best_len = 0;
    nicp_result best = { { 0 }, { 0 }, 0 };                               /* bestTransformation is default-constructed, :210 */
    float min_error = FLT_MAX;
    *error = 0.f;
    for (int rep = 0; rep < max_repetitions; rep++) {
        const int* perm = perms + (size_t)rep * size;
        memcpy(pb, before, sizeof(float) * 3 * (size_t)m);               /* ApplyPermutation keeps entries beyond the permutation, common.h:101-108 */
        memcpy(pa, after, sizeof(float) * 3 * (size_t)n);
        for (int i = 0; i < size; i++) {
            memcpy(pb + 3 * i, before + 3 * perm[i], sizeof(float) * 3);
            memcpy(pa + 3 * i, after + 3 * perm[i], sizeof(float) * 3);
        }
        nicp_result r;
        oracle_nicp_single(pb, m, pa, n, r.R, r.t, &r.approx);           /* :226 */
        *error = r.approx;
        if (approximation == 0) {
            *error = subcloud_error(sub, subcloud_n, &r, pa, n);         /* against the PERMUTED cloud, :231 */
            if (*error < min_error) {
                min_error = *error;
                best = r;
                if (min_error <= eps) {                                  /* :238-242 */
                    *repetitions = rep + 1;
                    goto done;
                }
            }
        } else {
            store_if_optimal(best_list, &best_len, &r, approximation == 2 ? 5 : 1);   /* :246-254 */
        }
    }
    *repetitions = max_repetitions;                                      /* :257 */
    if (approximation != 0) {                                            /* :261-279 */
        min_error = FLT_MAX;
        for (int i = 0; i < best_len; i++) {
            *error = subcloud_error(sub, subcloud_n, &best_list[i], after, n);
            if (*error < min_error) {
                min_error = *error;
                best = best_list[i];
                if (min_error <= eps) goto done;                         /* returns with *error = this error */
            }
        }
    }
    *error = min_error;                                                  /* :281 */
done:
    memcpy(rot9, best.R, sizeof best.R);
    memcpy(trans3, best.t, sizeof best.t);
    free(sub); free(pb); free(pa);
}
