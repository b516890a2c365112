// Internal launch interfaces between the C-ABI/driver layer (mislam_api.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mislam {

// ---------------------------------------------------------------------------------------------------------------
// K1 nearest-neighbour search (nn_kernel.hip)
// ---------------------------------------------------------------------------------------------------------------
constexpr int NN_TARGET_BLOCK = 16;       // T: targets per min-only block of K1 (chunk_len and target padding granule)
constexpr int NN_MAX_R = 8;               // sources per lane
constexpr int NN_SRC_PAD = 256 * NN_MAX_R;  // source arrays are padded to a multiple of this
constexpr unsigned long long KEY_INIT = 0xFFFFFFFFFFFFFFFFull;

struct NnLaunch {
    const float *sx, *sy, *sz;            // sources, SoA, n_pad floats each (n_pad % (256*R) == 0)
    int n, n_pad;
    const float *tx, *ty, *tz;            // targets, SoA, >= n_chunks*chunk_len floats each
    int chunk_len, n_chunks;              // chunk_len % NN_TARGET_BLOCK == 0
    int index_base;                       // global index of target 0 of this device's shard
    unsigned long long* keys;             // n packed (d2 bits << 32 | index) keys: KEY_INIT or a REAL candidate's key
    const int* done_flag;                 // device-side stop flag (may be null)
    int R;                                // 1, 2, 4 or 8
    int fma;
};
hipError_t nn_launch(const NnLaunch& a, hipStream_t stream);

// ---------------------------------------------------------------------------------------------------------------
// ICP iteration kernels (icp_kernels.hip)
// ---------------------------------------------------------------------------------------------------------------
constexpr int ICP_MOMENTS = 16;              // count, sum b (3), sum a (3), sum a b^T (9)
constexpr int ICP_ERRSUMS = 2;               // sum |a - b'|^2, kept pairs
constexpr int ICP_REDUCED_ROWS = 64;          // rows icp_rows_reduce leaves at most (= ICP_MAX_REDUCED_ROWS of icp_rows.hpp): one per lane of the solve kernel
constexpr int ICP_MAX_PARTIAL_BLOCKS = 512;   // 2 blocks per CU: enough loads in flight for the O(N) passes, few rows to reduce

// mirror of the public MI_STOP_* values (mi_slam.h) for device code
enum { MI_STOP_RUNNING_ = 0, MI_STOP_CONVERGED_ = 1, MI_STOP_MAX_ITERATIONS_ = 2, MI_STOP_NO_PAIRS_ = 3,
       MI_STOP_ERROR_INCREASED_ = 4, MI_STOP_TOLERANCE_ = 5, MI_STOP_SIGMA_ = 6 };

// Device-resident loop state.  The host only ever copies it back; every decision is taken on the device.
struct IcpState {
    float R[9];              // running rotation, column-major (glm::mat3)
    float t[3];              // running translation
    float prevR[9];
    float prevT[3];
    float Ri[9];             // last per-iteration solve
    float ti[3];
    float error;             // *error of the reference drivers
    float prev_error;
    int iterations;          // *iterations of the reference drivers
    int passes;              // loop bodies executed
    int done;
    int stop_reason;
    int pairs;               // correspondences kept in the last solve (this rank's share in the multi-GPU path)
    int err_pending;         // multi-GPU: err[] holds the last iteration's sums and its stop rule has not been evaluated yet
    double mom[ICP_MOMENTS];
    double err[ICP_ERRSUMS];
    // MI_SUM_CPU_SEQUENTIAL only: cpu-slam's own sequential fp32 running sums over the kept pairs in the caller's order
    float seq_sum_b[3];      // sum of the moving points        (GetCenterOfMass, common.cpp:281-284)
    float seq_sum_a[3];      // sum of their matched fixed points
    float seq_sum_err;       // sum of squared residuals         (GetMeanSquaredError, common.cpp:259-268)
    float seq_H[9];          // cpu-slam's cross-covariance of the kept pairs: sum fl32(a - ca) fl32(b - cb)^T with ITS centroids (round 6, icp_seq_cross_kernel)
};

struct IcpView {
    IcpState* state;
    const float *bx, *by, *bz;       // original `before`, SoA, n_pad
    float *cx, *cy, *cz;             // current (transformed) cloud, SoA, n_pad
    const float4* tgt4;              // this rank's target shard as float4 (gather-friendly), local index
    unsigned long long* keys;        // n packed keys
    int n, n_pad;
    int shard_lo, shard_hi;          // global target index range owned by this rank
    int filter_pairs;
    float max_distance_squared;
    int fma;                         // distance arithmetic used when re-arming keys with the previous match
    const int* inv_order;            // caller's index -> sorted slot (MI_SUM_CPU_SEQUENTIAL), else null
    float* resid;                    // per sorted slot: squared residual of the kept pair, +0 otherwise (same mode), else null
};

struct IcpRules {
    float eps;
    int max_iterations;
    int filter_pairs;
    int abort_on_increase;
    int m_total;                     // |after| over all ranks
    int seq_sums;                    // MI_SUM_CPU_SEQUENTIAL: the error comes from state->seq_sum_err
    int svd_ieee;                    // developer switch MISLAM_SVD_IEEE=1: K3 in IEEE divisions and roots instead of the refined hardware forms
};

constexpr int ICP_CHUNK_POINTS = 64;          // moving points per wave / per row of partial sums (icp_rows.hpp ICP_ROW_POINTS)
// stable LSD radix sort of (key, value) pairs on the low `bits` of the keys (radix_sort.hip); the *_in arrays are scratch afterwards
size_t radix_sort_temp_bytes(int n);
hipError_t radix_sort_pairs_u32(void* temp, unsigned int* keys_in, unsigned int* keys_out, int* vals_in, int* vals_out, int n, int bits,
                                hipStream_t s);
hipError_t fill_keys(unsigned long long* keys, int n, hipStream_t s);
// dst[i] (i < n_local) = the i-th point of the 64-point chunks rank, rank + world, rank + 2*world ... of src (n_all points); entries
// [n_local, n_pad) replicate the last one
hipError_t deal_chunks_soa(const float* sx, const float* sy, const float* sz, int n_all, int rank, int world, int n_local, int n_pad,
                           float* dx, float* dy, float* dz, hipStream_t s);
hipError_t aos_to_soa(const float* aos, int n, int n_pad, float* x, float* y, float* z, float4* packed, hipStream_t s);
hipError_t soa_to_aos(const float* x, const float* y, const float* z, int n, float* aos, hipStream_t s);
hipError_t unpack_keys(const unsigned long long* keys, const int* order, int n, int* idx, float* d2, hipStream_t s);
hipError_t pack_keys(const int* idx, const unsigned char* keep, int n, unsigned long long* keys, hipStream_t s);

int icp_reduce_blocks(int n);
// An iteration's sums are "rows" (icp_rows.hpp): one row of 18 partial sums -- 16 moments, 2 error sums -- per 64 moving points,
// whichever kernel produced them (the fused search of nn_grid.hip, or the two stand-alone kernels below).
int icp_row_count(int n);                  // rows of a cloud of n points
int icp_reduced_count(int nrows);          // rows left after icp_rows_reduce (<= 64)
hipError_t icp_moments_rows(const IcpView& v, double* rows, hipStream_t s);                    // K2: columns [0,16) of rows [0, row_count(n))
// K4+K5: cur = R*before + t for all n_pad entries, columns [16,18) of rows [0, row_count(n_pad)), keys re-armed:
// rearm 0 = leave keys, 1 = KEY_INIT, 2 = the previous match's key under the NEW transform (a real candidate: the next search
// starts from a tight bound)
hipError_t icp_transform_error_rows(const IcpView& v, double* rows, int rearm, hipStream_t s);
// Work order of the fused search (nn_grid.hip): order[position] = chunk; the rows-reduce kernel rewrites it every iteration from
// the flags the search left (far[chunk] != 0: its wave walked the box hierarchy), walking chunks first.  Scheduling only.
struct IcpSchedule {
    int* order;                            // nrows entries, always a permutation of the chunks
    unsigned char* far;                    // nrows flags
    unsigned long long* lanes;             // nrows lane masks (GridSearchArgs::far_lanes)
    int* counters;                         // 2 cursors, zeroed by the solve kernel
};
hipError_t icp_schedule_reset(const IcpSchedule& sched, int nrows, hipStream_t s);             // identity order, no flags
hipError_t icp_rows_reduce(const double* rows, int nrows, double* part, hipStream_t s, const IcpSchedule* sched = nullptr, bool all_rows = false);   // -> part[icp_reduced_count(nrows)][18]
constexpr int ICP_FUSED_SOLVE_MAX_ROWS = 2048;      // up to this many rows (131 072 moving points) rows reduce + solve are one launch of one workgroup
hipError_t icp_reduce_solve(IcpState* state, const double* rows, int nrows, int compose_mode, const IcpRules& rules, int mark_pending, hipStream_t s);
// reduced rows -> state->mom / state->err (which: 1 moments, 2 error sums, 3 both); the multi-GPU paths all-reduce them there
hipError_t icp_rows_to_state(IcpState* state, const double* part, int count, int which, hipStream_t s);
// K3 + K6, deferred: settles the PREVIOUS iteration's stop rule from the error sums (if state->err_pending), then -- unless it
// fired -- solves from the moments and composes.  part != null: sums = the reduced rows; null: already in state->mom / err.
// mark_pending: this iteration's own error will arrive with the next call (or with icp_finalize_pending).
hipError_t icp_solve_deferred(IcpState* state, const double* part, int count, int compose_mode, const IcpRules& rules, int mark_pending,
                              hipStream_t s, int* sched_counters = nullptr);
hipError_t icp_finalize_pending(IcpState* state, const double* part, int count, const IcpRules& rules, hipStream_t s);
hipError_t icp_mark_pending(IcpState* state, hipStream_t s);
// MI_SUM_CPU_SEQUENTIAL: cpu-slam's sequential fp32 running sums, reproduced bit for bit (one wave per sum)
hipError_t invert_order(const int* order, int n, int* inv, hipStream_t s);
hipError_t icp_seq_centroids(const IcpView& v, hipStream_t s);
hipError_t icp_seq_error(const IcpView& v, hipStream_t s);

// One per translation unit with kernels: loads that unit's code object (see the definitions).
hipError_t preload_nn_kernel();
hipError_t preload_nn_tree();
hipError_t preload_nn_grid();
hipError_t preload_icp_kernels();
hipError_t preload_cpd_kernels();
hipError_t preload_cpd_fgt();
hipError_t preload_nicp_api();
hipError_t preload_prepare_api();

}  // namespace mislam
