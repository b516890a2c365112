/* mi_slam.h -- C ABI of the MI355X-native point-set-registration core (libmislam.so).
 *
 * This is the drop-in boundary for the reference's GPU registration path.  A thin C++ adapter with the shape of the
 * reference's Common::SlamFunc (source/common/testrunner.h:7-8) binds these entry points; INTEGRATION.md shows the
 * adapter a maintainer would put where source/cuda-slam/gpumain.cpp:12-38 dispatches today.
 *
 * Conventions (all taken from the reference's own boundary):
 *   - clouds are contiguous AoS float xyz, 12 B per point, no padding: Common::Point<float>
 *     (source/common/point.h:6-64), the layout the reference memcpy's to the device as glm::vec3
 *     (source/cuda-slam/icpcuda.cu:71-72);
 *   - `before` is the moving cloud, `after` the fixed one; the result maps before onto after;
 *   - 4x4 results are COLUMN-MAJOR with the translation in column 3, byte-identical to glm::mat4 as built by
 *     Common::ConvertToTransformationMatrix (source/common/common.cpp:353-358) and to a default Eigen::Matrix4f;
 *   - host pointers in, host pointers out; the library owns every device allocation (context workspace);
 *   - every function returns MI_OK (0) or a negative error code instead of the reference's print-and-exit(1)
 *     (include/helper_cuda.h:567-573); mi_last_error() returns the message; nothing is printed unless verbose != 0.
 *
 * The hot path behind it is hand-written HIP for gfx950; there is no CPU fallback: if no HIP device is usable the
 * calls fail with MI_ERR_NO_DEVICE.
 */
#ifndef MI_SLAM_H
#define MI_SLAM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_SLAM_ABI_VERSION 4   /* 4: mi_profile_search_phases, mi_selftest_fail_loads, mi_runtime_info (additive: no signature of version 3 changed); 3: mi_icp_load_times, mi_cross_moments, mi_icp_auto_batch; 2: mi_cpd_params gained sigma2_mode; mi_dist_info, mi_source_share, mi_cpd_sigma_squared_mode, mi_profile_search_stats, mi_selftest_sort_pairs */

enum {
    MI_OK = 0,
    MI_ERR_INVALID_ARG = -1,
    MI_ERR_NO_DEVICE = -2,
    MI_ERR_HIP = -3,
    MI_ERR_RCCL = -4,
    MI_ERR_STATE = -5
};

/* Distance arithmetic of the nearest-neighbour search.  Both are IEEE fp32 on (dx,dy,dz) = target - source. */
enum {
    MI_DIST_CPU_ROUNDING = 0,   /* ((dx*dx + dy*dy) + dz*dz), every product and sum rounded: bit-identical to the x86-64
                                   cpu-slam build (source/common/point.h:48-50 via common.cpp:452) */
    MI_DIST_FMA = 1             /* fma(dz,dz, fma(dy,dy, dx*dx)): what nvcc's default contraction makes of
                                   GetDistanceSquared (source/cuda-slam/cudacommon.cu:51-55) */
};

/* Execution strategy of the nearest-neighbour search.  Every mode returns the SAME idx[] and d2[] bit for bit (strict '<',
 * lowest index on ties, same fp32 arithmetic); they differ only in how many candidate pairs are evaluated. */
enum {
    MI_NN_AUTO = 0,         /* cell grid for fixed clouds of >= MI_NN_INDEX_MIN_POINTS points per GPU, every pair below (measured crossover at N = M) */
    MI_NN_BRUTEFORCE = 1,   /* every pair, like FindCorrespondences (cudacommon.cu:57-77): N*M distance evaluations */
    MI_NN_TREE = 2,         /* exact search through a box hierarchy over the Morton-sorted fixed cloud */
    MI_NN_GRID = 3          /* exact search through a uniform cell grid over the fixed cloud; lanes the grid cannot serve cheaply
                               (no starting candidate, far outliers, crowded cells) walk the box hierarchy inside the same launch */
};
#define MI_NN_INDEX_MIN_POINTS 10000

/* What a multi-GPU context (mi_ctx_create_dist) splits across its ranks.  Same registration result either way.
 *   TARGET: rank r owns fixed points [M*r/W, M*(r+1)/W) and ALL moving points; one ncclAllReduce(ncclUint64, ncclMin) of the
 *           N packed (min-dist, argmin) keys per iteration, then owner-accumulated moments/error + one 64 x 18-double sum all-reduce.
 *           The every-pair search scales perfectly this way (its work is N*M/W per rank).
 *   SOURCE: every rank holds a replica of the fixed cloud (12 B/point: trivial at 288 GB) and 1/W of the moving points -- the
 *           64-point chunks of the moving cloud's Hilbert order dealt round-robin (rank r: chunks r, r+W, ...; contiguous
 *           slices [N*r/W, N*(r+1)/W) of the caller's order below 256*W points).  No per-point exchange at all, only one
 *           64 x 18-double sum all-reduce per iteration.  The indexed searches need this split to scale: their cost per moving
 *           point hardly depends on how many fixed points a rank holds.
 *   AUTO:   SOURCE when the search runs through an index (cell grid / box hierarchy), TARGET when it is the every-pair search. */
enum {
    MI_SHARD_AUTO = 0,
    MI_SHARD_TARGET = 1,
    MI_SHARD_SOURCE = 2
};

/* How the two centroids and the error of an ICP iteration are summed.
 *   EXACT:          fp64 two-stage sums (the default; correctly rounded to fp32 at the end).
 *   CPU_SEQUENTIAL: cpu-slam's own arithmetic, bit for bit: ONE sequential fp32 running sum per quantity over the kept
 *                   pairs in the caller's point order (GetCenterOfMass common.cpp:281-284, GetMeanSquaredError :259-268).
 *                   That sum is 1.3e-4 off the exact mean at 2e4 points and worse beyond, and it feeds both the translation
 *                   and the stop rule -- choose this mode to retrace cpu-slam's trajectory at sizes where its own summation
 *                   noise exceeds the 1e-4 parity budget.  Round 6: the mode also takes cpu-slam's own CROSS-COVARIANCE -- the kept pairs centred
 *                   in fp32 with those centroids (GetAlignedCloud, common.cpp:525-530) -- instead of the exact one: the two differ by ~1e-7, which
 *                   is nothing unless the matrix is rank-deficient (a first iteration that matches every moving point to the same two fixed points:
 *                   the exact matrix leaves R undetermined, cpu-slam's R is decided by that centring's rounding).  With it -- and the 3 x 3 SVD in IEEE
 *                   arithmetic, developer switch MISLAM_SVD_IEEE=1 -- the device retraces the CPU restatement of cpu-slam bit for bit on the
 *                   reference's convergence and sizes sets (tests/test_gpu_convergence_set.py).  Single-GPU contexts only; costs ~3.3 ms per million
 *                   points per iteration (a sequential fp32 sum cannot be re-associated). */
enum {
    MI_SUM_EXACT = 0,
    MI_SUM_CPU_SEQUENTIAL = 1
};

/* How the per-iteration solve (Ri, ti) is accumulated into the running transform. */
enum {
    MI_COMPOSE_CPU_ADDITIVE = 0,  /* R <- Ri*R ; t <- ti + t   -- source/cpu-slam/basicicp.cpp:43-44 (the oracle's rule) */
    MI_COMPOSE_EXACT = 1          /* T <- Ti*T (4x4)           -- source/cuda-slam/icpcuda.cu:35 */
};

/* Why a registration loop stopped. */
enum {
    MI_STOP_RUNNING = 0,
    MI_STOP_CONVERGED = 1,        /* error < eps                      basicicp.cpp:52 / icpcuda.cu:40 */
    MI_STOP_MAX_ITERATIONS = 2,   /* iterations reached the cap        basicicp.cpp:32 / icpcuda.cu:31 */
    MI_STOP_NO_PAIRS = 3,         /* no correspondence survived filter basicicp.cpp:36 */
    MI_STOP_ERROR_INCREASED = 4,  /* rollback rule                     icpcuda.cu:43-49 */
    MI_STOP_TOLERANCE = 5,        /* CPD: ntol <= tolerance            coherentpointdrift.cpp:106 */
    MI_STOP_SIGMA = 6             /* CPD: sigma^2 <= eps               coherentpointdrift.cpp:106 */
};

typedef struct mi_ctx mi_ctx;   /* opaque: device, stream, workspace, optional RCCL communicator */

/* ----------------------------------------------------------------------------------------------------------------
 * Context.  Hoists device selection, stream/event creation, workspace allocation and (multi-GPU) the communicator out
 * of the registration calls -- the reference allocates and frees everything per call (icpcuda.cu:21-29,56).
 * -------------------------------------------------------------------------------------------------------------- */
int mi_abi_version(void);
const char* mi_last_error(void);
int mi_device_count(int* count);

int mi_ctx_create(int device, mi_ctx** out);
/* Optional: load all device code now (~170 ms) instead of lazily at the first use of each kernel family, where it would stall
 * that registration call (first CPD call on the bunny clouds: 36 ms lazily, 3 ms after this).  For long-lived processes. */
int mi_ctx_preload(mi_ctx* ctx);

/* Multi-GPU, one process per GPU (no reference counterpart: the reference is single-GPU).  What the ranks split is
 * mi_icp_params.shard_mode (MI_SHARD_* above) for ICP -- with the fixed cloud sharded, the per-point packed (min-dist, argmin)
 * keys are combined with ONE ncclAllReduce(ncclUint64, ncclMin) over xGMI per iteration -- and the fixed cloud for CPD.
 * The library creates the communicator and issues every collective on its own stream.
 * unique_id is the 128-byte ncclUniqueId produced by mi_dist_unique_id() on rank 0 and shipped to the other ranks by
 * the caller's bootstrap (torch.distributed store, MPI, a socket ...). */
#define MI_UNIQUE_ID_BYTES 128
int mi_dist_unique_id(void* out_unique_id);
int mi_ctx_create_dist(int device, int rank, int world, const void* unique_id, mi_ctx** out);
int mi_ctx_rank(const mi_ctx* ctx, int* rank, int* world);
/* What the TRANSPORT says about this context (measurement hook: bench.py prints it as proof that the ranks really met):
 * *nranks / *rank as the RCCL communicator reports them (ncclCommCount / ncclCommUserRank; world / rank of an exchange context,
 * 1 / 0 of a single-GPU one), *ranks_seen the sum over all ranks of 2^rank carried through one all-reduce on the context's
 * stream -- bit r set = rank r took part.  Collective: every rank calls it.  Any output may be NULL. */
int mi_dist_info(mi_ctx* ctx, int* nranks, int* rank, unsigned long long* ranks_seen);

/* Which shared objects this library's HIP and RCCL calls actually bind to in THIS process, and their versions -- no device is touched
 * (dladdr on hipStreamSynchronize / ncclAllReduce, ncclGetVersion, hipRuntimeGetVersion).  A launcher that has imported another copy of the ROCm
 * runtime first (PyTorch bundles libamdhip64 / librccl under torch/lib with the same SONAMEs) decides what these resolve to: bench.py prints
 * them in its N > 1 line and refuses to start when two different RCCL objects of different versions are mapped (round 5).
 * Paths are copied into the caller's buffers (truncated to `cap` bytes, always NUL-terminated); any pointer may be NULL. */
int mi_runtime_info(char* hip_path, char* rccl_path, int cap, int* hip_runtime_version, int* rccl_version);

/* The same multi-GPU path over the CALLER's transport instead of RCCL (MPI between nodes, a test harness ...; no reference
 * counterpart either).  Wherever the RCCL context issues an all-reduce, this one drains its stream, copies the operand to
 * pinned host memory, calls `exchange` and copies the result back: `exchange` must combine host_buf[0 .. count) IN PLACE across
 * all `world` ranks -- element-wise unsigned 64-bit MIN for MI_EXCHANGE_MIN_U64, double SUM for MI_EXCHANGE_SUM_F64 -- and
 * return 0, or non-zero to fail the call with MI_ERR_RCCL.  Every rank makes the same sequence of calls.  Results are those
 * of the RCCL context up to the order in which the transport adds the doubles. */
enum { MI_EXCHANGE_MIN_U64 = 0, MI_EXCHANGE_SUM_F64 = 1 };
typedef int (*mi_exchange_fn)(void* user, void* host_buf, size_t count, int kind);
int mi_ctx_create_exchange(int device, int rank, int world, mi_exchange_fn exchange, void* user, mi_ctx** out);

/* Host-side pieces of the multi-GPU protocol (pure functions, usable without a device; the CPU tests drive them over gloo):
 *   mi_shard_range  the contiguous target range [lo, hi) rank `rank` of `world` owns: lo = M*rank/world, hi = M*(rank+1)/world;
 *   mi_source_share how many moving points rank `rank` of `world` works on under MI_SHARD_SOURCE (the 64-point chunks rank,
 *                   rank + world, ... of the cloud's Hilbert order; n*rank/world .. n*(rank+1)/world below 256*world points);
 *   mi_pack_key     the 64-bit key the search emits per source point: IEEE bits of d2 (d2 >= 0, so they order like the
 *                   value) in the high word, the GLOBAL target index in the low word -- an unsigned min over keys is the
 *                   (min distance, lowest index) rule of cudacommon.cu:68 / common.cpp:454, across chunks and across GPUs. */
int mi_shard_range(int m_total, int rank, int world, int* lo, int* hi);
int mi_source_share(int n_total, int rank, int world, int* count);
unsigned long long mi_pack_key(float d2, int global_index);
void mi_unpack_key(unsigned long long key, float* d2, int* global_index);
void mi_ctx_destroy(mi_ctx* ctx);
int mi_ctx_synchronize(mi_ctx* ctx);

/* ----------------------------------------------------------------------------------------------------------------
 * ICP -- replaces GetCudaIcpTransformationMatrix(before, after, eps, maxIterations, iterations, error)
 *        (source/cuda-slam/icpcuda.cuh:5-11, icpcuda.cu:60-76) and, through its modes, reproduces
 *        BasicICP::GetBasicICPTransformationMatrix (source/cpu-slam/basicicp.cpp:23-61), the parity oracle.
 * -------------------------------------------------------------------------------------------------------------- */
typedef struct {
    float eps;                   /* "convergence-epsilon"                (configparser.cpp:244, default 1e-3) */
    int   max_iterations;        /* "max-iterations", -1 = unbounded     (gpumain.cpp:14) */
    float max_distance_squared;  /* "max-distance-squared" (default 1000, configparser.cpp:207); used when filter_pairs */
    int   dist_mode;             /* MI_DIST_* */
    int   compose_mode;          /* MI_COMPOSE_* */
    int   filter_pairs;          /* 1: drop pairs with d2 >= max_distance_squared and average the error over the survivors
                                       (common.cpp:486-499, :267); 0: keep all and divide the error sum by |after|
                                       (cudacommon.cu:138-148) */
    int   abort_on_increase;     /* 1: stop and roll back when the error rises (icpcuda.cu:43-49); 0: cpu-slam never does */
    int   sync_every;            /* iterations enqueued between host checks of the device-side stop flag; 0 = auto.
                                    The stop rule itself is evaluated on the device after EVERY iteration, so the result
                                    does not depend on this value. */
    int   verbose;               /* 1: print "loop_nr %d, error: %f" lines like basicicp.cpp:50 at every host check */
    int   nn_mode;               /* MI_NN_*: how the correspondence search is carried out; the RESULT is identical in every mode */
    int   shard_mode;            /* MI_SHARD_*: what a multi-GPU context splits across ranks (ignored with one rank) */
    int   sum_mode;              /* MI_SUM_*: how the centroids and the error are summed */
    int   reserved[4];
} mi_icp_params;

/* Defaults = cpu-slam semantics (the parity oracle): CPU rounding, additive translation, filtered pairs, no abort. */
void mi_icp_params_default(mi_icp_params* p);
/* The GPU reference's own driver rules (icpcuda.cu:8-58): FMA distance, exact composition, no filter, abort+rollback. */
void mi_icp_params_cuda_slam(mi_icp_params* p);

/* One-call registration.  out_T: 16 floats column-major; iterations/error as the reference's out-params (non-null). */
int mi_icp_register(mi_ctx* ctx, const float* before_xyz, int n_before, const float* after_xyz, int n_after,
                    const mi_icp_params* params, float out_T[16], int* iterations, float* error);

/* Resident-data form of the same loop (what bench.py times: inputs already in HBM when the clock starts).
 *   mi_icp_load   uploads both clouds (AoS -> SoA on device), sizes the workspace, resets the state to identity;
 *   mi_icp_reset  resets the running transform / counters without re-uploading;
 *   mi_icp_run    enqueues up to `max_new_iterations` more iterations of the loop (fewer if a stop rule fires),
 *                 returns after they finished;
 *   mi_icp_result reads the current (T, iterations, error, stop reason). */
int mi_icp_load(mi_ctx* ctx, const float* before_xyz, int n_before, const float* after_xyz, int n_after,
                const mi_icp_params* params);
int mi_icp_reset(mi_ctx* ctx);
int mi_icp_run(mi_ctx* ctx, int max_new_iterations, int* iterations_done);
/* The iterations mi_icp_run enqueues between two host checks when sync_every = 0 -- a pure function of GLOBAL sizes (moving and
 * fixed points over all ranks, rank count, what is sharded, which search runs): every batch ends in a collective, so every rank
 * of a multi-GPU registration must pick the same number. */
int mi_icp_auto_batch(long long n_moving_total, long long m_fixed_total, int world, int source_sharded, int every_pair_search);
int mi_icp_result(mi_ctx* ctx, float out_T[16], int* iterations, float* error, int* stop_reason);

/* ----------------------------------------------------------------------------------------------------------------
 * Test-grade primitives (one per reference primitive on the path; host in / host out).
 * -------------------------------------------------------------------------------------------------------------- */

/* CUDACommon::GetCorrespondingPoints + FindCorrespondences (cudacommon.cu:272-289, :57-77) /
 * Common::GetCorrespondingPoints search part (common.cpp:441-478):
 * idx[i] = argmin_j |tgt[j] - src[i]|^2, strict '<', lowest index wins ties; d2[i] = that minimum (may be NULL). */
int mi_nn_search(mi_ctx* ctx, const float* src_xyz, int n, const float* tgt_xyz, int m, int dist_mode,
                 int* idx, float* d2);
/* Same, with an explicit MI_NN_* strategy (mi_nn_search uses MI_NN_AUTO). */
int mi_nn_search_ex(mi_ctx* ctx, const float* src_xyz, int n, const float* tgt_xyz, int m, int dist_mode, int nn_mode,
                    int* idx, float* d2);

/* The sums LeastSquaresSVD is built from (the cross-covariance product of cudacommon.cu:196-201 -- CuBlasMultiply -- and the two
 * centroid sums), over the kept pairs (src[i], tgt[idx[i]]):  out16 = { pairs, sum src (3), sum tgt (3), sum tgt_r * src_c (9, row-major
 * in r) }, fp64.  What the reference's own MultiplicationTest checks on its GEMM (cudacommon.cu:319-343: ones(3x100) * ones(100x3) is
 * 100 everywhere) is checked on these. */
int mi_cross_moments(mi_ctx* ctx, const float* src_xyz, int n, const float* tgt_xyz, int m, const int* idx,
                     const unsigned char* keep, double out16[16]);

/* CUDACommon::LeastSquaresSVD (cudacommon.cu:168-253) / Common::LeastSquaresSVD (common.cpp:517-552) on the pairs
 * (src[i], tgt[idx[i]]), i = 0..n-1, keeping pair i only if keep == NULL or keep[i] != 0.
 * out_R9 column-major (glm::mat3), out_t3 = centroid_tgt - R * centroid_src.  pairs_used may be NULL. */
int mi_kabsch(mi_ctx* ctx, const float* src_xyz, int n, const float* tgt_xyz, int m, const int* idx,
              const unsigned char* keep, float out_R9[9], float out_t3[3], int* pairs_used);

/* CUDACommon::TransformCloud + GetMeanSquaredError (cudacommon.cu:132-148) / common.cpp:219-224, :259-268:
 * out[i] = R*src[i] + t (glm operation order, no contraction); *mse = sum_i |tgt[idx[i]] - out[i]|^2 / denom over the
 * kept pairs, denom = number of kept pairs if divide_by_pairs else m.  out_xyz, idx, mse may be NULL. */
int mi_transform_mse(mi_ctx* ctx, const float* src_xyz, int n, const float R9[9], const float t3[3],
                     const float* tgt_xyz, int m, const int* idx, const unsigned char* keep, int divide_by_pairs,
                     float* out_xyz, float* mse);

/* ----------------------------------------------------------------------------------------------------------------
 * Rigid CPD, "approximation-type" none (exact Gaussian P), full or hybrid -- replaces GetCudaCpdTransformationMatrix
 * (source/cuda-slam/cpdcuda.cuh:5-17, cpdcuda.cu:302-386); oracle CoherentPointDrift::GetRigidCPDTransformationMatrix
 * (source/cpu-slam/coherentpointdrift.cpp:69-124).  CPD naming follows the reference: M = |before| (index k),
 * N = |after| (index x).
 * -------------------------------------------------------------------------------------------------------------- */
typedef struct {
    float eps;             /* "convergence-epsilon": loop runs while sigma^2 > eps      (coherentpointdrift.cpp:106) */
    float weight;          /* "cpd-weight" (default .3), clamped to [1e-6, 1-1e-6]      (:93-96) */
    int   const_scale;     /* "cpd-const-scale" (parser default false, configparser.cpp:240) */
    int   max_iterations;  /* loop runs while iterations < max_iterations; NB -1 therefore runs NO iteration (:106) */
    float tolerance;       /* "cpd-tolerance" (default 1e-3) */
    float sigma2_init;     /* > 0: use this initial sigma^2.  <= 0: computed on the device as the exact
                              sum_ij |b_i - a_j|^2 / (3MN) in fp64, or as sigma2_mode says.  cpu-slam's own value is a single
                              sequential fp32 running sum (coherentpointdrift.cpp:126-139) that saturates for M*N >~ 1e7
                              (3.604 instead of 12.943 on the bunny clouds): MI_SIGMA2_CPU_SEQUENTIAL reproduces it. */
    int   sync_every;      /* as in mi_icp_params */
    int   verbose;
    int   approximation;   /* MI_CPD_APPROX_*: "approximation-type" (configparser.cpp:221-230).  0 = the exact Gaussian P. */
    float fgt_ratio_of_far_field;    /* "fgt-ratio-of-far-field" e (default 10): cells farther than sqrt(e)*sigma are skipped */
    int   fgt_order_of_truncation;   /* "fgt-order-of-truncation" p (default 8): monomials of total degree < p, 1..16 */
    int   sigma2_mode;               /* MI_SIGMA2_*: how the initial sigma^2 is computed when sigma2_init <= 0 */
    int   estep_mode;                /* MI_ESTEP_*: the order the exact E-step's sums are added in (round 6; took one of the reserved words: same size) */
    int   reserved[3];
} mi_cpd_params;

/* The exact E-step's summation order.
 *   DEFAULT:        chunked partial sums over the whole chip (K7a / K7b).
 *   CPU_SEQUENTIAL: cpu-slam's own order -- each fixed point's affinities added one by one into one fp32 running sum, each moving point's P1 / PX one
 *                   fixed point at a time, value = p / denominator (coherentpointdrift.cpp:186-213).  A parity mode, ~5 x the default's time: cpu-slam's
 *                   running sums drop small terms whole, a one-sided error of ~2.5e-5 of sigma^2 per EM iteration that the default does not make and that
 *                   `cpd-const-scale: true` amplifies; with this mode the device retraces the CPU restatement through it.  Single-GPU contexts, exact P only. */
enum { MI_ESTEP_DEFAULT = 0, MI_ESTEP_CPU_SEQUENTIAL = 1 };

/* The initial sigma^2 = sum_ij |b_i - a_j|^2 / (3MN) (CalculateSigmaSquared, coherentpointdrift.cpp:126-139 / cpdcuda.cu:65-78).
 *   EXACT:          closed form from the clouds' sums, fp64 (what cuda-slam's thrust reduction approximates).
 *   CPU_SEQUENTIAL: cpu-slam's own arithmetic bit for bit -- ONE sequential fp32 running sum over all M*N squared distances, which
 *                   saturates once it dwarfs its terms (3.604 instead of 12.943 on the bunny clouds).  cpu-slam's whole EM
 *                   trajectory starts from that number, so this is the mode in which mi_cpd_register retraces cpu-slam without
 *                   being handed a constant.  A sequential sum cannot be re-associated, but inside one binade of the running sum
 *                   it IS an integer prefix sum of rne(term / ulp): computed binade by binade over the whole GPU, the blocks
 *                   at binade crossings and exact rounding ties term by term (cpd_kernels.hip) -- 3 ms for the bunny clouds'
 *                   2.2e8 pairs (round 2's one-wave retrace: 750 ms), once per registration; single-GPU contexts only. */
enum { MI_SIGMA2_EXACT = 0, MI_SIGMA2_CPU_SEQUENTIAL = 1 };

/* "approximation-type" of the reference (common/enumerators.h:18-23, coherentpointdrift.cpp:141-167):
 *   FULL    every E-step is the Fast Gauss Transform (common/fgt.cpp) and sigma^2 is clamped to >= 0.05;
 *   HYBRID  the FGT E-step while sigma^2 > 0.015 * sigma^2_init, then the exact P truncated below 1e-3 (the parser's default).
 * The reference runs the FGT on the CPU in both of its builds; here it runs on the device (K9).  With either
 * mode the loop is host-stepped (the cell count and the switch depend on sigma^2), one state read-back per iteration. */
enum { MI_CPD_APPROX_NONE = 0, MI_CPD_APPROX_FULL = 1, MI_CPD_APPROX_HYBRID = 2 };

void mi_cpd_params_default(mi_cpd_params* p);

/* out_sR_t: column-major 4x4 holding scale*R (cpdcuda.cu:360) and t; out_scale may be NULL.
 * On a multi-GPU context (mi_ctx_create_dist) every rank passes both clouds whole and keeps fixed points mi_shard_range(n_after);
 * one all-reduce of 24 doubles per EM iteration merges the M-step moments, every rank returns the same result.  The FGT modes
 * (approximation != MI_CPD_APPROX_NONE; hybrid is the reference parser's default, configparser.cpp:217) keep both clouds whole on every
 * rank -- they cluster whole clouds -- and split the E-step's QUERIES (round 6): the fixed points the first transform is evaluated at, the
 * moving points of the second, the truncated E-step's tiles; the per-point weights travel through an unsigned-minimum all-reduce (bit for
 * bit), the M-step's sums through the 24-double one: the single-GPU run's iteration counts, s R|t within 1e-6 of it.
 * Synchronous: when it returns, the context's stream is drained (round 5; the host checks inside only peek at the state, so up to one
 * iteration's worth of no-op launches or a prelaunched K-centre replay trails the copy that said "done" -- it is waited for here). */
int mi_cpd_register(mi_ctx* ctx, const float* before_xyz, int m_before, const float* after_xyz, int n_after,
                    const mi_cpd_params* params, float out_sR_t[16], float* out_scale, int* iterations, float* error);

/* CalculateSigmaSquared (cpdcuda.cu:65-78 / coherentpointdrift.cpp:126-139): the exact value, or with an explicit MI_SIGMA2_* mode. */
int mi_cpd_sigma_squared(mi_ctx* ctx, const float* before_xyz, int m, const float* after_xyz, int n, float* sigma2);
int mi_cpd_sigma_squared_mode(mi_ctx* ctx, const float* before_xyz, int m, const float* after_xyz, int n, int sigma2_mode, float* sigma2);

/* ComputePMatrix, no truncation (cpdcuda.cu:80-116 / coherentpointdrift.cpp:168-221):
 * p_xk = exp(-|x - y_k|^2 / (2 sigma^2)); den_x = sum_k p_xk + c; Pt1[x] = 1 - c/den_x; P1[k] = sum_x p_xk/den_x;
 * PX[k] = sum_x x * p_xk/den_x (row-major M x 3); *L = -sum_x log den_x + 1.5*N*log sigma^2. */
int mi_cpd_estep(mi_ctx* ctx, const float* y_xyz, int m, const float* x_xyz, int n, float constant, float sigma2,
                 float* p1, float* pt1, float* px, float* L);

/* ComputePMatrix with doTruncate (coherentpointdrift.cpp:168-221; the hybrid mode calls it with truncate = 1e-3, :166):
 * affinities with -|x - y_k|^2 / (2 sigma^2) < log(truncate) count as 0. */
int mi_cpd_estep_truncated(mi_ctx* ctx, const float* y_xyz, int m, const float* x_xyz, int n, float constant, float sigma2,
                           float truncate, float* p1, float* pt1, float* px, float* L);

/* ComputePMatrixWithFGT (common/cpdutils.cpp:19-77): the same four products through three Fast Gauss Transforms with
 * K = round(min(N, M, 50 + sigma2_init/sigma2)) cells, bandwidth sqrt(2 sigma2) and the outlier term taken at the CURRENT
 * sigma2.  Needs m, n >= 2 (the clustering starts from point 1, fgt.cpp:162). */
int mi_cpd_estep_fgt(mi_ctx* ctx, const float* y_xyz, int m, const float* x_xyz, int n, float weight, float sigma2,
                     float sigma2_init, float ratio_of_far_field, int order_of_truncation,
                     float* p1, float* pt1, float* px, float* L);

/* KCenter (common/fgt.cpp:152-212): greedy farthest-point clustering from point 1; cluster[i] in [0, K), centers_xyz[K][3] the
 * cell means.  Labels are bit-identical to the reference's (same arithmetic, same tie rules).  n >= 2, 1 <= K <= 65535. */
int mi_fgt_kcenter(mi_ctx* ctx, const float* cloud_xyz, int n, int K, float* centers_xyz, int* cluster);

/* The same clustering with a GUESS of the sweep's choices: guess[0 .. n_guess) = the points expected as centres 0, 1, ... (what an
 * earlier sweep of the same cloud under another similarity transform chose -- how rigid CPD's E-steps cluster the moving cloud: the
 * guess is replayed and checked for all points in parallel, and only what it does not cover, or gets wrong, is swept step by step).
 * The result is mi_fgt_kcenter's, bit for bit, whatever the guess.  picked[K] (may be NULL) receives the sweep's choices, *verified
 * (may be NULL) the number of leading entries of the guess that were the sweep's own (-1: no replay ran, n_guess < 2). */
int mi_fgt_kcenter_guided(mi_ctx* ctx, const float* cloud_xyz, int n, int K, const int* guess, int n_guess, float* centers_xyz,
                          int* cluster, int* picked, int* verified);

/* Host-only (no device needed): the monomial tables of truncation order p the FGT kernels use, pd = C(p+2,3) entries each in
 * the reference's graded monomial order (fgt.cpp:124-137): exponents packed a | b<<8 | c<<16, the constants 2^|alpha|/alpha!
 * of ComputeC_k (fgt.cpp:214-244), and each monomial's slot in the Horner traversal.  Any output may be NULL. */
int mi_fgt_tables(int order_of_truncation, unsigned int* mono, float* ck, int* horner_slot, int* pd);

/* MStep (cpdcuda.cu:172-300 / coherentpointdrift.cpp:223-277).  scale, sigma2 are in/out as in the reference. */
int mi_cpd_mstep(mi_ctx* ctx, const float* before_xyz, int m, const float* after_xyz, int n, const float* p1,
                 const float* pt1, const float* px, int const_scale, float out_R9[9], float out_t3[3], float* scale,
                 float* sigma2);

/* ----------------------------------------------------------------------------------------------------------------
 * Non-iterative registration, "method": "nicp" -- replaces GetCudaNicpTransformationMatrix (source/cuda-slam/nicpcuda.cuh:5-15,
 * nicpcuda.cu:70-184); oracle NonIterative::GetNonIterativeTransformationMatrix with the sequential policy
 * (source/cpu-slam/noniterative.cpp:204-282; the parallel policy races on the shared random generator, :84-101).
 * Each repetition aligns the principal axes of the two clouds, U_after * U_before^T, with the column signs an Eigen::JacobiSVD
 * of the randomly permuted 3 x N matrices would produce (they depend on the first three points only -- nicp_api.hip), and
 * the candidate with the smallest error wins:
 *   MI_CPD_APPROX_NONE    every candidate is scored on the comparison subcloud (exact nearest neighbours), early exit at eps;
 *   MI_CPD_APPROX_FULL    candidates ranked by the index-wise "approximated" error, the best one scored on the subcloud;
 *   MI_CPD_APPROX_HYBRID  the best five are scored on the subcloud (the parser's default).
 * The random choices stay with the caller, who draws them like the reference (std::mt19937 + std::shuffle,
 * GetRandomPermutationVector, common.cpp:554-560): order_heads[3*r .. 3*r+2] are the first three entries of repetition r's
 * permutation of 0..min(m,n)-1 (max_repetitions of them; -1 means 20), subcloud_idx the first subcloud_n entries of the
 * permutation GetSubcloud draws BEFORE them (common.cpp:25-37), or NULL with subcloud_n = m_before for the whole cloud.
 * -------------------------------------------------------------------------------------------------------------- */
typedef struct {
    float eps;              /* "convergence-epsilon" */
    int   max_repetitions;  /* "nicp-iterations" (default 32); -1 = 20 (noniterative.cpp:207-208) */
    int   approximation;    /* MI_CPD_APPROX_* ("approximation-type") */
    int   verbose;
    int   reserved[4];
} mi_nicp_params;

void mi_nicp_params_default(mi_nicp_params* p);

/* out_T: column-major 4x4 (R | t).  *repetitions and *error as the reference leaves them. */
int mi_nicp_register(mi_ctx* ctx, const float* before_xyz, int m_before, const float* after_xyz, int n_after,
                     const mi_nicp_params* params, const int* order_heads, const int* subcloud_idx, int subcloud_n,
                     float out_T[16], int* repetitions, float* error);

/* ----------------------------------------------------------------------------------------------------------------
 * Input stage: what Common::GetCloudsFromConfig (source/common/common.cpp:134-210) does to ONE cloud between
 * LoadCloud and the registration call, on the device and with the reference's arithmetic bit for bit:
 *   GetSubcloud (common.cpp:25-37) -> NormalizeCloud to "cloud-spread" (:81-95; the centre of mass is the reference's
 *   sequential fp32 accumulate, :281-284) -> std::shuffle (:166-167) -> AddNoiseToCloud (:97-119) -> AddOutliersToCloud
 *   (:121-132) -> GetTransformedCloud (:219-224, the `after` cloud only).
 * The random outcomes are the caller's, drawn like the reference draws them (mt19937 + std::shuffle for the index vectors,
 * rand() for the unit draws; host/cloud_io.cpp does it):
 *   subcloud_idx   first subcloud_n entries of GetSubcloud's permutation of 0..n_raw-1, or NULL (whole cloud);
 *   shuffle_idx    the shuffle as a gather order -- std::shuffle of an iota of the cloud's size with the same generator:
 *                  prepared row i = normalised row shuffle_idx[i] -- or NULL;
 *   noise_rows     ascending rows of the shuffled cloud that AddNoiseToCloud's flag permutation marks, noise_unit[3q..3q+2]
 *                  the three rand()/RAND_MAX draws of the q-th of them (x, y, z), in the order the reference consumes them;
 *   outlier_unit   three draws per appended outlier.
 * out_xyz holds (cloud size + n_outliers) points; *out_n receives that count.
 * -------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int   has_spread;       /* "cloud-spread" present: normalise to `spread` */
    float spread;
    float noise_intensity;  /* "noise-intensity-before" / "-after" */
    int   has_transform;    /* apply rotation (column-major, like glm::mat3) and translation: p -> R p + t */
    float rotation[9];
    float translation[3];
    int   reserved[4];
} mi_prepare_params;

void mi_prepare_params_default(mi_prepare_params* p);

int mi_prepare_cloud(mi_ctx* ctx, const float* raw_xyz, int n_raw, const int* subcloud_idx, int subcloud_n, const int* shuffle_idx,
                     const int* noise_rows, const float* noise_unit, int n_noise, const float* outlier_unit, int n_outliers,
                     const mi_prepare_params* params, float* out_xyz, int* out_n);

/* ----------------------------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py): per-kernel HIP-event timing on the context's own stream.
 * -------------------------------------------------------------------------------------------------------------- */
enum {
    MI_KERNEL_NN = 0,          /* K1 brute-force nearest-neighbour search */
    MI_KERNEL_MOMENTS = 1,     /* K2 fused gather + centroid/cross-covariance partial sums */
    MI_KERNEL_SOLVE = 2,       /* K3 reduce + 3x3 SVD + compose */
    MI_KERNEL_TRANSFORM = 3,   /* K4/K5 transform + error partial sums + key reset */
    MI_KERNEL_FINALIZE = 4,    /* K6 error reduce + stop rule */
    MI_KERNEL_ALLREDUCE = 5,   /* C1 RCCL packed-min all-reduce (multi-GPU) */
    MI_KERNEL_CPD_DENOM = 6,   /* K7a column sums (den, Pt1, L) */
    MI_KERNEL_CPD_CONTRACT = 7,/* K7b P~.[X|1] contraction (MFMA) */
    MI_KERNEL_CPD_MSTEP = 8,   /* K8 weighted moments + solve */
    MI_KERNEL_CPD_FGT = 9,     /* K9 Fast-Gauss-Transform E-step (clustering, model, predict) */
    MI_KERNEL_COUNT = 10
};
int mi_profile_enable(mi_ctx* ctx, int enable);
/* Which kernels get events while profiling is enabled: bit k = MI_KERNEL_k (default: all).  Two event records around every
 * launch of a 5-kernel iteration cost a few per cent of a millisecond-sized step; bench.py times only the search in its timed
 * region and the rest in a separate short run. */
int mi_profile_select(mi_ctx* ctx, unsigned int kernel_mask);
int mi_profile_reset(mi_ctx* ctx);
/* total_ms = sum of event-timed durations of that kernel since the last reset; launches = how many. */
int mi_profile_get(mi_ctx* ctx, int kernel, double* total_ms, long long* launches);
/* Where the last mi_icp_load (or the load inside mi_icp_register) spent its host wall time, in ms -- what the reference's own
 * timing of a whole SlamFunc call includes besides the iterations (testrunner.cpp:54-56, doc/documentation.tex:397):
 *   out[0] workspace (device allocations)          out[1] moving cloud: upload + AoS -> SoA
 *   out[2] moving cloud: Hilbert order + permute    out[3] fixed cloud: upload + AoS -> SoA
 *   out[4] box hierarchy over the fixed cloud       out[5] cell grid over the fixed cloud
 *   out[6] state reset                              out[7] the whole load
 * While profiling is enabled (mi_profile_enable) the stream is drained after every stage, so the parts are attributable (and no
 * longer overlap); otherwise only out[7] is meaningful. */
#define MI_LOAD_STAGES 8
int mi_icp_load_times(mi_ctx* ctx, double out_ms[MI_LOAD_STAGES]);
/* Work counters of the cell-grid search (MI_NN_GRID), summed over its launches while enabled: out[0] candidates tested in the
 * grid, out[1] cell rows scanned, out[2] moving points that went on to the box hierarchy, out[3] moving points searched, out[4]
 * hierarchy nodes and out[5] leaves visited by out[6] walking waves, out[7] nodes + leaves of the longest single walk (a
 * maximum, not a sum).
 * enable != 0 starts counting (and zeroes the counters), 0 stops; out may be NULL.  Costs four atomics per moving point while on. */
int mi_profile_search_stats(mi_ctx* ctx, int enable, unsigned long long out[8]);
/* The same counting build's LOOP TRIP COUNTS since counting was switched on (call it BEFORE mi_profile_search_stats(ctx, 0, ..) switches it off), summed over
 * waves -- what tools/isa_budget.py multiplies the kernel's static per-phase instruction counts by (the measured instruction budget of DESIGN section 4):
 * out[0] waves, [1] waves that ran the grid scan, [2] walk-only waves; of the scanning waves: [3] nearest-block batches, [4] of them with their trips dealt
 * over the wave, [5] deal passes, [6] iterations of the deal's write loop, [7] lockstep trips; [8] rounds of leftover rows dealt out, [9] of them with their
 * trips dealt, [10] deal passes, [11] write-loop iterations, [12] lockstep trips, [13] leftover batches of the four-rows-per-lane form, [14] waves that had
 * leftover rows; of the walking waves: [15] leaves that held a candidate at or below some lane's best, [16] leaf children looked at, [17] votes for the nearest
 * child, [18] pops of a pending child, [19] leaves whose sequential (index-settling) offers ran.  Measurement hook: no reference counterpart. */
int mi_profile_search_phases(mi_ctx* ctx, unsigned long long out[20]);
/* Self-test of the library's own device radix sort (the Hilbert ordering of the index build): sorts the n (key, value) pairs in
 * place, stable, ascending by the low `bits` (10, 20 or 30) of the keys.  Host arrays; test use only. */
int mi_selftest_sort_pairs(mi_ctx* ctx, unsigned int* keys, int* values, int n, int bits);
/* Fault injection for the library's own tests: the next n index builds of this context fail with MI_ERR_INVALID_ARG behind the fixed cloud's upload (the
 * early-return path of mi_icp_load).  Only ever by this explicit call -- round 5 read it from the environment, where a stray variable could have failed
 * loads in production.  n = 0 disarms. */
int mi_selftest_fail_loads(mi_ctx* ctx, int n);
/* Name of the correspondence-search kernel (MI_KERNEL_NN) a search of n_moving points against m_fixed_local fixed points runs
 * with this nn_mode and the current settings -- the name a rocprofv3 kernel trace shows (static string). */
const char* mi_nn_kernel_name(const mi_ctx* ctx, int n_moving, int m_fixed_local, int nn_mode);

#ifdef __cplusplus
}
#endif
#endif /* MI_SLAM_H */
