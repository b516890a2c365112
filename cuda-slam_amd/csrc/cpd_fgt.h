// Internal launch interfaces of the Fast-Gauss-Transform E-step (cpd_fgt.hip): the reference's approximation-type
// "full" / "hybrid" (source/common/fgt.cpp, source/common/cpdutils.cpp:19-77), restated for the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace mislam {

constexpr int FGT_MAX_ORDER = 16;          // p: monomials of total degree < p; exponents are packed in 8 bits each
constexpr int FGT_MAX_CLUSTERS = 65535;    // K <= this (the member-list sort's scratch is sized for it)
constexpr int FGT_GRID_SWEEP_MIN_POINTS = 65536;   // above this the K-centre sweep runs one grid-wide launch per step
constexpr int FGT_GRID_SWEEP_BLOCKS = 1024;
constexpr size_t FGT_SWEEP_SCRATCH_BYTES = (size_t)(FGT_GRID_SWEEP_BLOCKS + 1) * 20;   // current centre + per-workgroup arg-max records

// Monomial tables of one truncation order p, device resident (built on the host, cpd_api.hip).  Index t is the reference's
// graded order (fgt.cpp:124-137): degree by degree, x-power descending, then y-power descending.
struct FgtTables {
    const unsigned int* mono;   // [pd]  a | b << 8 | c << 16
    const float* ck;            // [pd]  2^|alpha| / alpha!, rounded step by step as ComputeC_k does (fgt.cpp:214-244)
    const int* hpos;            // [pd]  position of monomial t in the Horner traversal the predict kernel reads
    int p, pd;
};

// One cloud clustered into K cells (KCenter, fgt.cpp:152-212)
struct FgtClusters {
    const float *x, *y, *z;     // the cloud, SoA
    int n, K;
    int k_done;                 // > 0: dist/indx hold a finished sweep of this cloud with k_done < K centres -- resume it
    float* dist;                // [n]   squared distance to the nearest centre so far (kept: the sweep can be resumed)
    int* indx;                  // [n]   cluster of each point
    int* memb;                  // [n]   point ids grouped by cluster, ascending inside a cluster
    int* off;                   // [K+1] cluster k owns memb[off[k] .. off[k+1])
    float* xc;                  // [K][3] cluster means
    void* sweep_scratch;        // FGT_SWEEP_SCRATCH_BYTES, used by the grid-wide sweep (may be null for small clouds)
    int* picked;                // [K]   id of the point the sweep chose as centre k (written by every sweep)
    // guess > 1 (and k_done == 0): picked[0 .. guess) holds a GUESS of the sweep's choices -- the previous E-step's, for a cloud that
    // only underwent a similarity transform since -- to be replayed and checked in parallel instead of swept step by step
    // (fgt_replay_kernel).  Needs the two buffers below; the result does not depend on the guess.
    int guess;
    unsigned long long* replay_partial;   // [fgt_replay_limit(guess, K)][fgt_replay_waves(n)]
    int* replay_state;                    // [1]  -> the number of leading steps of the guess that were verified
    int replay_done;                      // the replay of this guess is on the stream already (fgt_replay_prelaunch): fgt_cluster only resumes from its verdict
    int centers_in_model;                 // round 5: fgt_cluster leaves the cluster means to the model build that follows (fgt_model(..., centers = true)): one launch less
    int coop_sweep;                       // round 5: 1 = a sweep of at least 16 steps over more than 16 384 points runs on several workgroups in one cooperative launch
                                          // (fgt_kcenter_coop_kernel), 2 = every sweep of such a cloud (tests), 0 = never (rounds 1-4)
    int lists_in_model;                   // round 5 (with centers_in_model, clouds of at most 32 768 points): ... and the member lists too -- every cell's workgroup lists its own
                                          // members (the same memb / off as the sort's three launches leave, written by fgt_model instead)
};
constexpr int FGT_REPLAY_MAX_CENTRES = 4000;       // centres a replay stages in LDS (16 bytes each)
int fgt_replay_waves(int n);
int fgt_replay_limit(int guess, int K);
// the replay + check of c.guess alone (what fgt_cluster starts with): lets a caller put it on the stream before it knows K -- rigid CPD does,
// behind the transform of the moving cloud, while the host still waits for sigma^2 (K never shrinks below the guess there)
hipError_t fgt_replay_prelaunch(const FgtClusters& c, hipStream_t s);

size_t fgt_sort_temp_bytes(int n);     // scratch fgt_cluster needs for its member-list sort
// K-centre clustering + member lists + cluster means; everything a model build needs
hipError_t fgt_cluster(const FgtClusters& c, void* sort_temp, size_t sort_temp_bytes, hipStream_t s);
// coefficients B[k][hpos][w] = C_k * sum_{i in cluster k} weight_w(i) exp(-|dx|^2) dx^alpha, dx = (pt - xc_k) / sigma.
// w4 == nullptr: one weight set of ones (W = 1); else four: (w4.x, w4.y, w4.z, w4.w) (W = 4).
// centers: compute the cluster means first, in the same kernel (what fgt_centers_kernel would have left in c.xc, the same bits), for a
// clustering made with centers_in_model.
// part / Z (round 5, big cells): Z = fgt_model_splits(n, K, pd) > 1 and part = Z x K x pd x W floats of scratch: the cells' member lists are split over Z
// workgroups each, their partial sums added in z order (the means, if asked for, in a launch of their own).
constexpr int FGT_MODEL_MAX_SPLITS = 16;
int fgt_model_splits(int n, int K, int pd);
// whether a cloud of n points in K cells is small enough for the model build to list every cell's members itself (FgtClusters::lists_in_model): a cloud that
// is takes ONE workgroup per cell on every E-step -- the caller passes Z = 1 for it whether or not this E-step re-clusters (the path, and with it
// the order of a cell's sums, must follow from n, K and pd alone)
bool fgt_lists_rule(int n, int K);
hipError_t fgt_model(const FgtClusters& c, const float4* w4, float sigma, const FgtTables& t, float* B, hipStream_t s, bool centers = false,
                     float* part = nullptr, int Z = 1);
// v[split][w][i] = sum_{k in split} [ |dy|^2 <= e ] exp(-|dy|^2) sum_alpha B[k][alpha][w] dy^alpha, dy = (q_i - xc_k) / sigma
// (fgt.cpp:88-150); the S = fgt_predict_splits(nq, K) partial sums are added in split order by the post kernels
int fgt_predict_splits(int nq, int K);
hipError_t fgt_predict(const float* qx, const float* qy, const float* qz, int nq, const float* xc, const float* B, int K, int W,
                       float sigma, float e_param, const FgtTables& t, int S, float* v, hipStream_t s);
// Kt1 -> 1/denominator, Pt1 and the four weight sets of the second transform (cpdutils.cpp:45-52, :79-99)
// (xpartials != null: + the M-step's x-sums of what it has just produced, in nblocks rows of CPD_XSUMS -- the terms, the grouping and hence the
// bits of cpd_xsums_kernel launched with the same nblocks; round 5: two launches less per FGT iteration with the k-sums below)
hipError_t fgt_post_kt1(const float* kt1_parts, int S, const float* ax, const float* ay, const float* az, int n, float ndi, float* pt1,
                        float4* xw4, hipStream_t s, double* xpartials = nullptr, int nblocks = 0);
// v[S][4][m] -> P1[m], PX[m][3]   (kpartials != null: + the M-step's k-sums, as cpd_ksums_kernel; b = the original moving cloud)
hipError_t fgt_post_px(const float* v_parts, int S, int m, float* p1, float* px, hipStream_t s, double* kpartials = nullptr, int nblocks = 0,
                       const float* bx = nullptr, const float* by = nullptr, const float* bz = nullptr);

}  // namespace mislam
