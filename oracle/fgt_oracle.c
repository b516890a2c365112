/* TEST INFRASTRUCTURE ONLY -- never linked into, imported by or shipped with the product path.
 *
 * Plain-C restatement of the reference's Fast-Gauss-Transform E-step and of its full / hybrid CPD drivers
 * (approximation-type "full" / "hybrid"): source/common/fgt.cpp, source/common/cpdutils.cpp and
 * source/cpu-slam/coherentpointdrift.cpp.  Pinned against the reference itself (oracle/_ref, the same sources compiled in
 * place) by tests/test_oracle_vs_ref.py and against the fixtures under tests/golden/ generated from it. */
#define _GNU_SOURCE 1
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "slam_oracle.h"

/* nchoosek: source/common/fgt.cpp:304-321 */
int oracle_fgt_nchoosek(int n, int k)
{
    int n_k = n - k;
    int r = 1;
    if (k < n_k) { k = n_k; n_k = n - k; }
    for (int i = 1; i <= n_k; i++) { r *= (++k); r /= i; }
    return r;
}

int oracle_fgt_pd(int p) { return oracle_fgt_nchoosek(p + 3 - 1, 3); }      /* fgt.cpp:69 */

static float len2(float x, float y, float z) { return x * x + y * y + z * z; }   /* Point::LengthSquared, point.h:49-51 */

/* KCenter: source/common/fgt.cpp:152-212.  Greedy farthest-point clustering that starts from point 1 (not 0, not random),
 * takes the FIRST maximum of the distance array as the next centre (std::max_element) and moves a point only on a strictly
 * smaller distance; the K centres returned are the means of the clusters (sequential float sums in index order). */
void oracle_fgt_kcenter(const float* cloud, int n, int K, float* xc, int* indx)
{
    float* dist = (float*)malloc(sizeof(float) * (size_t)n);
    int* boxsz = (int*)calloc((size_t)K, sizeof(int));
    int center = 1;                                               /* :162 */
    for (int i = 0; i < n; i++) {                                 /* :170-175 */
        dist[i] = len2(cloud[3 * i] - cloud[3 * center], cloud[3 * i + 1] - cloud[3 * center + 1], cloud[3 * i + 2] - cloud[3 * center + 2]);
        indx[i] = 0;
    }
    for (int i = 1; i < K; i++) {                                 /* :177-193 */
        center = 0;
        for (int j = 1; j < n; j++) if (dist[j] > dist[center]) center = j;
        for (int j = 0; j < n; j++) {
            const float d = len2(cloud[3 * j] - cloud[3 * center], cloud[3 * j + 1] - cloud[3 * center + 1], cloud[3 * j + 2] - cloud[3 * center + 2]);
            if (d < dist[j]) { dist[j] = d; indx[j] = i; }
        }
    }
    for (int i = 0; i < 3 * K; i++) xc[i] = 0.f;                  /* :195-199 */
    for (int i = 0; i < n; i++) {                                 /* :201-205 */
        boxsz[indx[i]]++;
        for (int d = 0; d < 3; d++) xc[3 * indx[i] + d] += cloud[3 * i + d];
    }
    for (int i = 0; i < K; i++) {                                 /* :207-210: xc *= 1.0f / count (0 members -> inf * 0 = NaN, as there) */
        const float inv = 1.0f / (float)boxsz[i];
        for (int d = 0; d < 3; d++) xc[3 * i + d] *= inv;
    }
    free(dist); free(boxsz);
}

/* The graded monomial recursion shared by ComputeC_k (:214-244), ComputeA_k (:246-302) and ComputeFGTPredict (:88-150):
 * prods[0] = seed, then degree by degree prods[t] = v[i] * prods[j] for j over the previous degree's monomials that do not
 * contain a lower-numbered coordinate.  parent/coord (optional) receive the recursion's structure. */
static void monomials(const float v[3], float seed, int p, float* prods)
{
    int heads[3] = { 0, 0, 0 };
    prods[0] = seed;
    for (int k = 1, t = 1, tail = 1; k < p; k++, tail = t)
        for (int i = 0; i < 3; i++) {
            const int head = heads[i];
            heads[i] = t;
            for (int j = head; j < tail; j++, t++) prods[t] = v[i] * prods[j];
        }
}

/* ComputeC_k: fgt.cpp:214-244 -- C[alpha] = 2^|alpha| / alpha!, built by the same recursion in double steps narrowed to float. */
void oracle_fgt_ck(int p, float* C_k)
{
    const int pd = oracle_fgt_pd(p);
    int* cinds = (int*)malloc(sizeof(int) * (size_t)pd);
    int heads[4] = { 0, 0, 0, INT_MAX };
    cinds[0] = 0;
    C_k[0] = 1.0f;
    for (int k = 1, t = 1, tail = 1; k < p; k++, tail = t)
        for (int i = 0; i < 3; i++) {
            const int head = heads[i];
            heads[i] = t;
            for (int j = head; j < tail; j++, t++) {
                cinds[t] = (j < heads[i + 1]) ? cinds[j] + 1 : 1;
                C_k[t] = (float)(2.0 * C_k[j]);
                C_k[t] = (float)(C_k[t] / (double)cinds[t]);
            }
        }
    free(cinds);
}

/* ComputeFGTModel: fgt.cpp:63-86.  ak is pd x K column-major (column k = cluster k), as the reference's Eigen matrix. */
void oracle_fgt_model(const float* cloud, int n, const float* weights, float sigma, int K, int p, float* xc, float* ak)
{
    const int pd = oracle_fgt_pd(p);
    int* indx = (int*)malloc(sizeof(int) * (size_t)n);
    float* C_k = (float*)malloc(sizeof(float) * (size_t)pd);
    float* prods = (float*)malloc(sizeof(float) * (size_t)pd);
    memset(ak, 0, sizeof(float) * (size_t)pd * (size_t)K);
    oracle_fgt_kcenter(cloud, n, K, xc, indx);
    oracle_fgt_ck(p, C_k);
    const float inv = 1.0f / sigma;                               /* ComputeA_k :260 */
    for (int i = 0; i < n; i++) {                                 /* :268-297 */
        const int k = indx[i];
        float dx[3];
        for (int d = 0; d < 3; d++) dx[d] = (cloud[3 * i + d] - xc[3 * k + d]) * inv;
        monomials(dx, expf(-len2(dx[0], dx[1], dx[2])), p, prods);
        for (int a = 0; a < pd; a++) ak[(size_t)k * pd + a] += weights[i] * prods[a];
    }
    for (int k = 0; k < K; k++)                                   /* :299-305 */
        for (int a = 0; a < pd; a++) ak[(size_t)k * pd + a] *= C_k[a];
    free(indx); free(C_k); free(prods);
}

/* ComputeFGTPredict: fgt.cpp:88-150 */
void oracle_fgt_predict(const float* cloud, int n, const float* xc, const float* ak, float sigma, float e_param, int K, int p, float* v)
{
    const int pd = oracle_fgt_pd(p);
    float* prods = (float*)malloc(sizeof(float) * (size_t)pd);
    const float inv = 1.0f / sigma;
    for (int m = 0; m < n; m++) {
        float cell = 0.f;
        for (int k = 0; k < K; k++) {
            float dy[3];
            for (int d = 0; d < 3; d++) dy[d] = (cloud[3 * m + d] - xc[3 * k + d]) * inv;
            const float sum = len2(dy[0], dy[1], dy[2]);
            if (sum > e_param) continue;                          /* :120 */
            monomials(dy, expf(-sum), p, prods);
            for (int a = 0; a < pd; a++) cell += ak[(size_t)k * pd + a] * prods[a];   /* :141-144 */
        }
        v[m] = cell;
    }
    free(prods);
}

/* K of the FGT E-step: cpdutils.cpp:36 */
int oracle_cpd_fgt_clusters(int m, int n, float sigma_squared, float sigma_squared_init)
{
    float k = 50.0f + sigma_squared_init / sigma_squared;
    if ((float)n < k) k = (float)n;
    if ((float)m < k) k = (float)m;
    return (int)roundf(k);
}

/* ndi of the FGT E-step: cpdutils.cpp:46 -- the outlier constant from the CURRENT sigma^2 (the exact path keeps the initial one) */
float oracle_cpd_fgt_ndi(float sigma_squared, float weight, int m, int n)
{
    return (float)((pow(2 * M_PI * sigma_squared, (double)(3.f * 0.5f)) * weight * m) / (double)((1 - weight) * n));
}

/* ComputePMatrixWithFGT: source/common/cpdutils.cpp:19-77 */
void oracle_cpd_estep_fgt(const float* transformed, int m, const float* after, int n, float weight, float sigma_squared,
                          float sigma_squared_init, float ratio_of_far_field, float order_of_truncation,
                          float* p1, float* pt1, float* px, float* L)
{
    const float hsigma = sqrtf(2.0f * sigma_squared);             /* :31 */
    const float e_param = ratio_of_far_field;
    const int K = oracle_cpd_fgt_clusters(m, n, sigma_squared, sigma_squared_init);
    const int p = (int)order_of_truncation;                       /* :37 */
    const int pd = oracle_fgt_pd(p);
    float* xc = (float*)malloc(sizeof(float) * 3 * (size_t)K);
    float* ak = (float*)malloc(sizeof(float) * (size_t)pd * (size_t)K);
    float* ones = (float*)malloc(sizeof(float) * (size_t)m);
    float* kt1 = (float*)malloc(sizeof(float) * (size_t)n);
    float* inv = (float*)malloc(sizeof(float) * (size_t)n);
    float* w = (float*)malloc(sizeof(float) * (size_t)n);
    float* col = (float*)malloc(sizeof(float) * (size_t)m);
    for (int k = 0; k < m; k++) ones[k] = 1.0f;
    oracle_fgt_model(transformed, m, ones, hsigma, K, p, xc, ak);                 /* :42 */
    oracle_fgt_predict(after, n, xc, ak, hsigma, e_param, K, p, kt1);             /* :43 */
    const float ndi = oracle_cpd_fgt_ndi(sigma_squared, weight, m, n);            /* :45 */
    for (int x = 0; x < n; x++) inv[x] = 1.0f / (kt1[x] + ndi);                   /* :49 */
    for (int x = 0; x < n; x++) pt1[x] = 1.0f - ndi * inv[x];                     /* CalculatePt1 :79-88 */
    oracle_fgt_model(after, n, inv, hsigma, K, p, xc, ak);                        /* :54 */
    oracle_fgt_predict(transformed, m, xc, ak, hsigma, e_param, K, p, p1);        /* :55 */
    for (int d = 0; d < 3; d++) {                                                 /* :59-66 */
        for (int x = 0; x < n; x++) w[x] = after[3 * x + d] * inv[x];             /* CalculateWeightsForPX :90-99 */
        oracle_fgt_model(after, n, w, hsigma, K, p, xc, ak);
        oracle_fgt_predict(transformed, m, xc, ak, hsigma, e_param, K, p, col);
        for (int k = 0; k < m; k++) px[3 * k + d] = col[k];
    }
    float acc = 0.0f;                                                             /* :69-71 */
    for (int x = 0; x < n; x++) acc += logf(kt1[x] + ndi);
    float error = -acc;
    error += (float)(3 * n) * logf(sigma_squared) / 2.0f;                         /* :72 (int * int * float / float) */
    *L = error;
    free(xc); free(ak); free(ones); free(kt1); free(inv); free(w); free(col);
}

/* ComputePMatrix with doTruncate: source/cpu-slam/coherentpointdrift.cpp:168-221 (the hybrid mode calls it with 1e-3, :166) */
void oracle_cpd_estep_truncated(const float* transformed, int m, const float* after, int n, float constant, float sigma_squared,
                                float truncate, float* p1, float* pt1, float* px, float* L)
{
    const float multiplier = -0.5f / sigma_squared;
    float* p = (float*)malloc(sizeof(float) * (size_t)m);
    memset(p1, 0, sizeof(float) * (size_t)m);
    memset(px, 0, sizeof(float) * 3 * (size_t)m);
    float error = 0.f;
    truncate = logf(truncate);                                    /* :182-183 */
    for (int x = 0; x < n; x++) {
        const float ax = after[3 * x], ay = after[3 * x + 1], az = after[3 * x + 2];
        float denominator = 0.f;
        for (int k = 0; k < m; k++) {
            const float dx = ax - transformed[3 * k], dy = ay - transformed[3 * k + 1], dz = az - transformed[3 * k + 2];
            const float index = multiplier * (dx * dx + dy * dy + dz * dz);
            if (index < truncate) p[k] = 0.0f;                    /* :193-196 */
            else { const float value = expf(index); p[k] = value; denominator += value; }
        }
        denominator += constant;
        pt1[x] = 1.0f - constant / denominator;
        for (int k = 0; k < m; k++)
            if (p[k] != 0.0f) {
                const float value = p[k] / denominator;
                p1[k] += value;
                px[3 * k] += ax * value; px[3 * k + 1] += ay * value; px[3 * k + 2] += az * value;
            }
        error -= logf(denominator);
    }
    error += (float)((size_t)3 * (size_t)n) * logf(sigma_squared) / 2.0f;
    *L = error;
    free(p);
}

/* GetRigidCPDTransformationMatrix with any approximation type: coherentpointdrift.cpp:69-124 + ComputePMatrixFast :141-167.
 * approximation: 0 none, 1 full, 2 hybrid (enumerators.h:18-23). */
void oracle_cpd_approx(const float* before, int m, const float* after, int n, const oracle_cpd_params* p, int approximation,
                       float ratio_of_far_field, float order_of_truncation, float rot9[9], float trans3[3], int* iterations,
                       float* error, float* trace, int trace_cap, int* trace_len)
{
    *iterations = 0;
    *error = 1e5f;
    float R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    float t[3] = { 0, 0, 0 };
    float scale = 1.0f;
    float sigma2 = oracle_cpd_sigma_squared(before, m, after, n);
    const float sigma2_init = sigma2;
    float weight = p->weight;
    if (weight <= 0.0f) weight = 1e-6f;
    if (weight >= 1.0f) weight = 1.0f - 1e-6f;
    const float constant = oracle_cpd_constant(sigma2, weight, m, n);
    float ntol = p->tolerance + 10.0f;
    float l = 0.0f;
    float* cur = (float*)malloc(sizeof(float) * 3 * (size_t)m);
    float* p1 = (float*)malloc(sizeof(float) * (size_t)m);
    float* pt1 = (float*)malloc(sizeof(float) * (size_t)n);
    float* px = (float*)malloc(sizeof(float) * 3 * (size_t)m);
    memcpy(cur, before, sizeof(float) * 3 * (size_t)m);
    int tl = 0;
    while (*iterations < p->max_iterations && ntol > p->tolerance && sigma2 > p->eps) {
        float L = 0.f;
        int used_fgt = 0;
        if (approximation == 0) {
            oracle_cpd_estep(cur, m, after, n, constant, sigma2, p1, pt1, px, &L);
        } else if (approximation == 1) {                          /* :152-157: sigma^2 is clamped (and stays clamped for the caller) */
            if (sigma2 < 0.05) sigma2 = 0.05;
            oracle_cpd_estep_fgt(cur, m, after, n, weight, sigma2, sigma2_init, ratio_of_far_field, order_of_truncation, p1, pt1, px, &L);
            used_fgt = 1;
        } else {                                                  /* :158-164 */
            if (sigma2 > 0.015 * sigma2_init) {
                oracle_cpd_estep_fgt(cur, m, after, n, weight, sigma2, sigma2_init, ratio_of_far_field, order_of_truncation, p1, pt1, px, &L);
                used_fgt = 1;
            } else {
                oracle_cpd_estep_truncated(cur, m, after, n, constant, sigma2, 1e-3f, p1, pt1, px, &L);
            }
        }
        ntol = fabsf((L - l) / L);
        l = L;
        oracle_cpd_mstep(before, m, after, n, p1, pt1, px, p->const_scale, R, t, &scale, &sigma2);
        oracle_transform_cloud(before, m, R, t, scale, 1, cur);
        *error = sigma2;
        (*iterations)++;
        if (trace && tl < trace_cap) {
            float* tr = trace + 17 * (size_t)tl;
            tr[0] = sigma2; tr[1] = L; tr[2] = ntol; tr[3] = scale; memcpy(tr + 4, R, sizeof R); memcpy(tr + 13, t, sizeof t);
            tr[16] = (float)used_fgt;
        }
        tl++;
    }
    for (int i = 0; i < 9; i++) rot9[i] = scale * R[i];
    memcpy(trans3, t, sizeof t);
    if (trace_len) *trace_len = tl;
    free(cur); free(p1); free(pt1); free(px);
}
