// Internal launch interfaces of the rigid-CPD kernels (cpd_kernels.hip).  Naming follows the reference's CPD code
// (source/common/cpdutils.cpp:28-29): M = |before| = moving cloud y_k (index k), N = |after| = fixed cloud x (index x).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace mislam {

constexpr int CPD_T = 8;              // scalar-load block of the broadcast stream
constexpr int CPD_XSUMS = 8;          // sum log den, sum pt1*a (3), sum pt1*|a|^2, 3 spare
constexpr int CPD_KSUMS = 16;         // Np, sum p1*b (3), sum b_r*px_c (9, row-major in r), sum p1*|b|^2, 2 spare
constexpr int CPD_INIT_SUMS = 8;      // sum a (3), sum |a|^2, sum b (3), sum |b|^2
constexpr int CPD_MAX_CHUNKS = 256;

struct CpdState {
    float R[9];            // rotation, column-major
    float t[3];
    float scale;
    float sigma2;
    float sigma2_init;
    float constant;        // c of coherentpointdrift.cpp:98 (fixed from sigma2_init)
    float L;               // Probabilities::error of the last E-step
    float l_prev;
    float ntol;
    float error;
    float Np;
    int iterations;
    int done;
    int stop_reason;
    double xs[CPD_XSUMS];
    double ks[CPD_KSUMS];
    double init[CPD_INIT_SUMS];
};

struct CpdRules {
    float eps, weight, tolerance;
    int const_scale, max_iterations;
    int m, n;
    int svd_ieee;                    // developer switch MISLAM_SVD_IEEE=1 (svd3.hpp kabsch_rotation)
};

struct CpdView {
    CpdState* state;
    // moving cloud: original b (SoA) and current y = s*R*b + t (SoA), m_pad entries each
    const float *bx, *by, *bz;
    float *yx, *yy, *yz;
    int m;
    // fixed cloud a (SoA), n_pad entries
    const float *ax, *ay, *az;
    int n;
    // E-step products
    float* den_part;       // [x_chunks_of_k][n]   partial sum_k p_xk
    float4* xw4;           // [n]  (w*ax, w*ay, w*az, w), w = 1/den_x
    float* pt1;            // [n]
    float* p1_part;        // [chunks][m]
    float* px_part;        // [chunks][3][m]  (SoA per component)
    float* p1;             // [m]
    float* px;             // [m][3] row-major, the reference's px.row(k)
    int k_chunks, k_chunk_len;   // K7a: chunks over k
    int x_chunks, x_chunk_len;   // K7b: chunks over x
    // hybrid mode's truncated kernel (coherentpointdrift.cpp:182-196): affinities whose exponent is < trunc_log count as 0
    int truncate;
    float trunc_log;
};

// K7t: the truncated E-step of the hybrid mode, culled by tile boxes along a space-filling curve (cpd_trunc.hip)
constexpr int CPD_TRUNC_TILE = 64;        // points per tile = one wave
constexpr int CPD_TRUNC_GROUP = 16;       // points per group: what is tested and staged on the side that is streamed (16 is built into the kernels' lane maps: not a knob)
constexpr int CPD_TRUNC_MAX_BLOCKS = 4096;   // workgroups (= rows of M-step partial sums) of its two kernels
struct CpdTruncView {
    CpdState* state;
    // fixed cloud in curve order (padded to whole tiles), its tile boxes (component-major: [6][tiles]), sorted slot -> caller's index
    const float *ax, *ay, *az;
    const float *abox, *agroup;           // boxes of its tiles of 64 / groups of 16
    const int* a_order;
    int n;
    // moving cloud: CURRENT positions in the curve order of the original cloud (padded to whole tiles), boxes of this E-step, slot -> caller's index
    const float *yx, *yy, *yz;
    const float *ybox, *ygroup;
    const int* b_order;
    int m;
    const float *bx, *by, *bz;            // the original moving cloud, caller's order (M-step k-sums)
    float4* xw4;                          // [n] curve order: (w*ax, w*ay, w*az, w), w = 1/den_x
    float4* xw4_caller;                   // the same records in the caller's order (CpdView::xw4)
    float *pt1, *p1, *px;                 // caller's order, as CpdView
    float trunc_log;
    // Round 6, multi-rank contexts (the whole clouds: 0 .. tiles): the denominators kernel works on the FIXED cloud's tiles [a_tile_lo, a_tile_hi) of this rank, the
    // contraction on the MOVING cloud's tiles [y_tile_lo, y_tile_hi) -- each against the whole other cloud, so every per-point value is the single-GPU run's to the
    // bit; between the two the ranks exchange xw4 (cpd_api.hip)
    int a_tile_lo, a_tile_hi, y_tile_lo, y_tile_hi;
};
// out = in[order] (padded to whole tiles with copies of the last point) and the tiles' boxes; state != null: nothing once it says done
hipError_t cpd_trunc_gather(const float* x, const float* y, const float* z, const int* order, int n, float* ox, float* oy, float* oz,
                            float* tile_box, float* group_box, const CpdState* state, hipStream_t s);
hipError_t cpd_trunc_denominators(const CpdTruncView& v, double* xpartials, int nblocks, hipStream_t s);   // den, Pt1, xw4 + the M-step's x-sums
hipError_t cpd_trunc_contract(const CpdTruncView& v, double* kpartials, int nblocks, hipStream_t s);       // P1, PX + the M-step's k-sums

// MI_ESTEP_CPU_SEQUENTIAL: den (into den_part[0 .. n)), Pt1, xw4, then P1 / PX, every sum in cpu-slam's index order (parity mode; cpd_kernels.hip)
hipError_t cpd_estep_sequential(const CpdView& v, hipStream_t s);
hipError_t cpd_init_sums(const CpdView& v, double* partials, int nblocks, hipStream_t s);
// sigma2_override > 0: use it; sigma2_from_state: use the value cpd_sigma2_sequential left in state->sigma2_init; else the exact
// closed form from the sums
hipError_t cpd_init_state(CpdState* state, const double* partials, int nblocks, const CpdRules& rules, float sigma2_override,
                          int sigma2_from_state, hipStream_t s);
// cpu-slam's own sigma^2_0 bit for bit (coherentpointdrift.cpp:126-139): ONE sequential fp32 running sum over all m*n squared
// distances, before-major -- into state->sigma2_init.  Computed binade by binade as integer prefix sums (cpd_kernels.hip); drains
// the stream a few dozen times.  scratch: cpd_sigma2_scratch_bytes() of device memory; host_pinned: pinned host memory for two ints.
size_t cpd_sigma2_scratch_bytes();
hipError_t cpd_sigma2_sequential(const CpdView& v, hipStream_t s, void* scratch, int* host_pinned);
hipError_t cpd_denominators(const CpdView& v, hipStream_t s);                                  // K7a
// (xpartials / kpartials != null: the post kernels also leave the M-step's x-sums / k-sums, as cpd_xsums / cpd_ksums would, in nblocks rows)
hipError_t cpd_post_denominators(const CpdView& v, hipStream_t s, double* xpartials = nullptr, int nblocks = 0);                             //   den, w, Pt1, xw4
hipError_t cpd_contract(const CpdView& v, int use_mfma, hipStream_t s);                        // K7b
hipError_t cpd_post_contract(const CpdView& v, hipStream_t s, double* kpartials = nullptr, int nblocks = 0);                                 //   P1, PX from chunk partials
hipError_t cpd_xsums(const CpdView& v, double* partials, int nblocks, hipStream_t s);          // K8 part 1
hipError_t cpd_ksums(const CpdView& v, double* partials, int nblocks, hipStream_t s);          // K8 part 2
// multi-GPU: a rank's own sums into state->xs/ks resp. state->init (all-reduced there); cpd_solve / cpd_init_state then take
// them from the state block when called with zero partial rows
hipError_t cpd_reduce_sums(CpdState* state, const double* xpart, int nxb, const double* kpart, int nkb, hipStream_t s);
hipError_t cpd_reduce_init(CpdState* state, const double* partials, int nblocks, hipStream_t s);
hipError_t cpd_solve(CpdState* state, const double* xpart, int nxb, const double* kpart, int nkb, const CpdRules& rules,
                     int update_loop_state, hipStream_t s);                                    // K8 solve (+ EM bookkeeping)
hipError_t cpd_transform(const CpdView& v, int m_pad, hipStream_t s);                          // y = s*R*b + t
hipError_t cpd_set_sigma2(CpdState* state, float sigma2, hipStream_t s);

}  // namespace mislam
