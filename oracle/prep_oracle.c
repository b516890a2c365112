/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's input stage, the checker for mi_prepare_cloud.
 * Nothing under cuda-slam_amd/ may include, link or call this file.
 *
 * Follows Common::GetCloudsFromConfig (source/common/common.cpp:134-210) for ONE cloud, stage by stage:
 *   GetSubcloud :25-37, NormalizeCloud :81-95 (GetCenterOfMass :281-284, GetAlignedCloud :327-333, CalculateCloudSpread :57-79),
 *   std::shuffle :166-167, AddNoiseToCloud :97-119, AddOutliersToCloud :121-132, GetTransformedCloud :219-224 /
 *   TransformPoint :45-49, Tests::GetRandomFloat testutils.cpp:7-11.
 * The random outcomes (index vectors, unit draws) are inputs: the generators behind them are library-specific.
 * Pinned against the reference itself (oracle/_ref, tests/test_prepare_oracle.py) and the committed fixture it produced
 * (tests/golden/bunny_prepare.npz, oracle/make_golden_prepare.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "slam_oracle.h"

static void bounds(const float* pts, int n, float lo[3], float hi[3])
{
    for (int k = 0; k < 3; k++) { lo[k] = pts[k]; hi[k] = pts[k]; }
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            const float v = pts[3 * (size_t)i + k];
            if (v < lo[k]) lo[k] = v;
            if (v > hi[k]) hi[k] = v;
        }
}

static float spread_of(const float* pts, int n)     /* CalculateCloudSpread: the largest axis-aligned span */
{
    float lo[3], hi[3];
    bounds(pts, n, lo, hi);
    float s = hi[0] - lo[0];
    if (hi[1] - lo[1] > s) s = hi[1] - lo[1];
    if (hi[2] - lo[2] > s) s = hi[2] - lo[2];
    return s;
}

int oracle_prepare_cloud(const float* raw, int n_raw, const int* subcloud_idx, int subcloud_n, const int* shuffle_idx,
                         const int* noise_rows, const float* noise_unit, int n_noise, float noise_intensity,
                         const float* outlier_unit, int n_outliers, int has_spread, float spread, const float* rot9_colmajor,
                         const float* trans3, float* out)
{
    const int n = subcloud_idx ? subcloud_n : n_raw;
    float* cur = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++)                                             /* GetSubcloud: cloud[perm[i]], i < size */
        memcpy(cur + 3 * (size_t)i, raw + 3 * (size_t)(subcloud_idx ? subcloud_idx[i] : i), 3 * sizeof(float));

    if (has_spread) {                                                       /* NormalizeCloud */
        float c[3] = {0.f, 0.f, 0.f};
        for (int i = 0; i < n; i++)                                         /* std::accumulate of Point_f: three running fp32 sums */
            for (int k = 0; k < 3; k++) c[k] = c[k] + cur[3 * (size_t)i + k];
        for (int k = 0; k < 3; k++) c[k] = c[k] / (float)n;
        float* al = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) al[3 * (size_t)i + k] = cur[3 * (size_t)i + k] - c[k];
        const float mx = spread_of(al, n);
        if (!(fabs((double)mx) < 1e-15)) {
            const float scale = spread / mx;
            for (int i = 0; i < n; i++)
                for (int k = 0; k < 3; k++) cur[3 * (size_t)i + k] = al[3 * (size_t)i + k] * scale - c[k] * -1.f;
        }
        free(al);
    }

    for (int i = 0; i < n; i++)                                             /* std::shuffle as a gather */
        memcpy(out + 3 * (size_t)i, cur + 3 * (size_t)(shuffle_idx ? shuffle_idx[i] : i), 3 * sizeof(float));
    free(cur);

    if (n_noise > 0) {                                                      /* AddNoiseToCloud */
        const float reach = spread_of(out, n) * noise_intensity;
        const float mn = -reach, range = reach - mn;
        for (int q = 0; q < n_noise; q++)
            for (int k = 0; k < 3; k++) {
                float* p = out + 3 * (size_t)noise_rows[q] + k;
                *p = *p + (noise_unit[3 * (size_t)q + k] * range + mn);
            }
    }
    if (n_outliers > 0) {                                                   /* AddOutliersToCloud */
        float lo[3], hi[3];
        bounds(out, n, lo, hi);
        for (int q = 0; q < n_outliers; q++)
            for (int k = 0; k < 3; k++) {
                const float range = hi[k] - lo[k];
                out[3 * (size_t)(n + q) + k] = outlier_unit[3 * (size_t)q + k] * range + lo[k];
            }
    }
    if (rot9_colmajor) {                                                    /* TransformPoint: glm mat3 * vec3, then + t */
        const float* r = rot9_colmajor;
        for (int i = 0; i < n + n_outliers; i++) {
            float* p = out + 3 * (size_t)i;
            const float x = p[0], y = p[1], z = p[2];
            for (int k = 0; k < 3; k++) p[k] = (r[k] * x + r[3 + k] * y + r[6 + k] * z) + trans3[k];
        }
    }
    return n + n_outliers;
}
