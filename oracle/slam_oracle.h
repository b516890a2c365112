/* TEST INFRASTRUCTURE ONLY -- see slam_oracle.c.  Plain-C restatement of the cpu-slam hot path. */
#ifndef SLAM_ORACLE_H
#define SLAM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* All clouds are contiguous AoS float xyz (12 B/point, source/common/point.h:61-63).
 * All 3x3 matrices cross this interface COLUMN-MAJOR (m[3*col+row]) like glm::mat3. */

enum { ORACLE_DIST_CPU = 0,   /* ((dx*dx + dy*dy) + dz*dz), every op rounded: x86-64 cpu-slam build (point.h:48-50) */
       ORACLE_DIST_FMA = 1 }; /* fmaf(dz,dz, fmaf(dy,dy, dx*dx)): what nvcc's default -fmad=true makes of cudacommon.cu:51-55 */

enum { ORACLE_COMPOSE_CPU_ADDITIVE = 0, /* R <- Ri*R ; t <- ti + t      (cpu-slam/basicicp.cpp:43-44) */
       ORACLE_COMPOSE_EXACT = 1 };      /* T <- Ti*T (4x4 product)      (cuda-slam/icpcuda.cu:35)     */

void  oracle_nn_search(const float* before, int n, const float* after, int m, int threads, int dist_mode,
                       int* idx, float* d2);
int   oracle_filter_pairs(const float* d2, int n, float max_distance_squared, int* idx_before);
void  oracle_center_of_mass(const float* pts, int k, float out3[3]);
void  oracle_jacobi_svd3(const float a_rowmajor[9], float u_rowmajor[9], float s[3], float v_rowmajor[9]);
void  oracle_kabsch_from_h(const float h_rowmajor[9], const float center_before[3], const float center_after[3],
                           float rot9[9], float trans3[3], float sing[3], float* det_sign);
void  oracle_least_squares_svd(const float* before_pts, const float* after_pts, int k, float rot9[9], float trans3[3]);
void  oracle_transform_cloud(const float* cloud, int n, const float rot9[9], const float trans3[3], float scale,
                             int use_scale, float* out);
float oracle_mse_indexed(const float* before, const float* after, const int* idx_before, const int* idx_after, int k);

typedef struct {
    float eps;
    float max_distance_squared;
    int   max_iterations;     /* -1 = unbounded */
    int   threads;            /* <=0: all hardware threads (common.cpp:443) */
    int   dist_mode;
    int   compose_mode;
    int   abort_on_increase;  /* cuda-slam/icpcuda.cu:43-49 rollback rule (0 = cpu-slam behaviour) */
    int   filter_pairs;       /* 1 = cpu-slam: drop pairs with d2 >= max_distance_squared, average over survivors;
                                 0 = cuda-slam: keep all, divide MSE by |after| (cudacommon.cu:147) */
} oracle_icp_params;

/* trace (optional, may be NULL): per executed loop body 14 floats = error, kept pairs, R(9, column-major), t(3). */
void  oracle_icp(const float* before, int n, const float* after, int m, const oracle_icp_params* p,
                 float rot9[9], float trans3[3], int* iterations, float* error, float* trace, int trace_cap, int* trace_len);

float oracle_cpd_sigma_squared(const float* before, int m, const float* after, int n);
float oracle_cpd_constant(float sigma_squared, float weight, int m_before, int n_after);
void  oracle_cpd_estep(const float* transformed, int m, const float* after, int n, float constant, float sigma_squared,
                       float* p1, float* pt1, float* px, float* L);
void  oracle_cpd_mstep(const float* before, int m, const float* after, int n, const float* p1, const float* pt1,
                       const float* px, int const_scale, float rot9[9], float trans3[3], float* scale, float* sigma_squared);

typedef struct {
    float eps;
    float weight;
    int   const_scale;
    int   max_iterations;
    float tolerance;
} oracle_cpd_params;

/* trace (optional): per EM iteration 16 floats = sigma2, L, ntol, scale, R(9), t(3). rot9 returns scale*R. */
void  oracle_cpd(const float* before, int m, const float* after, int n, const oracle_cpd_params* p,
                 float rot9[9], float trans3[3], int* iterations, float* error, float* trace, int trace_cap, int* trace_len);

/* ---- Fast Gauss Transform E-step and the full / hybrid CPD drivers (oracle/fgt_oracle.c) ---- */
int   oracle_fgt_nchoosek(int n, int k);
int   oracle_fgt_pd(int p);
void  oracle_fgt_kcenter(const float* cloud, int n, int K, float* xc, int* indx);
void  oracle_fgt_ck(int p, float* C_k);
void  oracle_fgt_model(const float* cloud, int n, const float* weights, float sigma, int K, int p, float* xc, float* ak);
void  oracle_fgt_predict(const float* cloud, int n, const float* xc, const float* ak, float sigma, float e_param, int K, int p,
                         float* v);
int   oracle_cpd_fgt_clusters(int m, int n, float sigma_squared, float sigma_squared_init);
float oracle_cpd_fgt_ndi(float sigma_squared, float weight, int m, int n);
void  oracle_cpd_estep_fgt(const float* transformed, int m, const float* after, int n, float weight, float sigma_squared,
                           float sigma_squared_init, float ratio_of_far_field, float order_of_truncation,
                           float* p1, float* pt1, float* px, float* L);
void  oracle_cpd_estep_truncated(const float* transformed, int m, const float* after, int n, float constant, float sigma_squared,
                                 float truncate, float* p1, float* pt1, float* px, float* L);
/* approximation: 0 none, 1 full, 2 hybrid.  trace: per EM iteration 17 floats = sigma2, L, ntol, scale, R(9), t(3), used_fgt. */
void  oracle_cpd_approx(const float* before, int m, const float* after, int n, const oracle_cpd_params* p, int approximation,
                        float ratio_of_far_field, float order_of_truncation, float rot9[9], float trans3[3], int* iterations,
                        float* error, float* trace, int trace_cap, int* trace_len);

/* ---- non-iterative registration, "method": "nicp" (oracle/nicp_oracle.c) ---- */
void  oracle_nicp_single(const float* before, int m, const float* after, int n, float rot9[9], float trans3[3],
                         float* approximated_error);
/* perms: max_repetitions permutations of 0..min(m,n)-1; subcloud_idx NULL = the whole cloud.  approximation: 0 none, 1 full, 2 hybrid. */
void  oracle_nicp(const float* before, int m, const float* after, int n, float eps, int max_repetitions, int approximation,
                  const int* subcloud_idx, int subcloud_n, const int* perms, float rot9[9], float trans3[3], int* repetitions,
                  float* error);

/* ---- input stage, Common::GetCloudsFromConfig for one cloud (oracle/prep_oracle.c) ---- */
/* NULL index vectors / zero counts / rot9_colmajor NULL: stage absent.  out holds (cloud size + n_outliers) points; returns that count. */
int   oracle_prepare_cloud(const float* raw, int n_raw, const int* subcloud_idx, int subcloud_n, const int* shuffle_idx,
                           const int* noise_rows, const float* noise_unit, int n_noise, float noise_intensity,
                           const float* outlier_unit, int n_outliers, int has_spread, float spread, const float* rot9_colmajor,
                           const float* trans3, float* out);

#ifdef __cplusplus
}
#endif
#endif
