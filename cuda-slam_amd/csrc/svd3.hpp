// K3 building block: 3x3 singular value decomposition + Kabsch rotation, usable from one GPU lane (and from host code).
//
// The reference solves the 3x3 problem with cusolverDnSgesvd plus five blocking memcpys per iteration
// (source/cuda-slam/cudacommon.cu:203-224) on the GPU and with Eigen::JacobiSVD on the CPU (source/common/common.cpp:531).
// Here it is a two-sided Jacobi iteration in registers, the scheme Eigen's JacobiSVD documents for square input
// (scale by the largest |a_ij|; sweep the pairs (1,0), (2,0), (2,1); symmetrise each 2x2 block with a left rotation, then
// diagonalise it with a Jacobi rotation; stop when a sweep rotates nothing; flip signs so the diagonal is positive; sort
// descending), so the result tracks the CPU oracle to rounding.  R = U diag(1,1,det(U V^T)) V^T as in
// common.cpp:541-545 / cudacommon.cu:236-238.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

namespace mislam {

struct Mat3 {
    float a[3][3];   // a[row][col]
};

// The solve runs on ONE lane, a chain of dependent instructions: what it costs is the length of that chain (round 3: ~3 600
// instructions, among them 37 IEEE divisions at ~10 instructions each and 9 correctly rounded square roots).  FAST (device code
// only) takes the quotients and roots that feed the rotations through the hardware approximations refined by fused multiply-adds:
// v_rcp_f32 + two residual steps (6 instructions), v_rsq_f32 + one (4), the root as x * rsq + one (5).  Eigen's sweep order, skip
// thresholds, sign fix and sort are untouched; the result moves by rounding, not in kind (bunny: 1.4e-5 from cpu-slam after 39
// iterations, IEEE 1.2e-5).  FAST = false is the IEEE form (host code, and the non-iterative method, whose sign pattern is compared
// with the reference's permutation by permutation).
#ifndef MISLAM_SVD_REFINE
#define MISLAM_SVD_REFINE 1
#endif
// (What matters is not the size of the rounding but its BIAS: a rotation whose c^2 + s^2 errs to one side scales U, V and so R, and
// the ICP driver COMPOSES 40 of those -- measured on the bunny clouds: Newton steps written with separate multiplies and adds land
// 4e-4 from cpu-slam after 39 iterations, the raw hardware forms 2.3e-5, IEEE 1.2e-5.  Hence residuals through fused multiply-adds:
// each refinement step then ends in ONE rounding of an almost exact value, like the IEEE operation it replaces.)
template <bool FAST>
struct SvdMath {
    static __host__ __device__ __forceinline__ float div(float a, float b)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        if (FAST) {
            float r = __builtin_amdgcn_rcpf(b);
#if MISLAM_SVD_REFINE
            r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.f), r, r);           // r <- r + r (1 - b r)
            const float q = a * r;
            const float v = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);  // q <- q + r (a - b q): the quotient to the last bit but for rare ties
#else
            const float v = a * r;
#endif
            return v;                // (NaN where the quotient overflows: see kabsch_rotation)
        }
#endif
        return a / b;
    }
    static __host__ __device__ __forceinline__ float rsqrt(float x)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        if (FAST) {
            float y = __builtin_amdgcn_rsqf(x);
#if MISLAM_SVD_REFINE
            y = __builtin_fmaf(0.5f * y, __builtin_fmaf(-x * y, y, 1.f), y);  // y <- y + (y / 2) (1 - x y^2)
#endif
            return y;                // (x = inf: 0 * inf in the residual -> NaN, see kabsch_rotation)
        }
#endif
        return 1.f / sqrtf(x);
    }
    static __host__ __device__ __forceinline__ float sqrt(float x)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        if (FAST) {
#if MISLAM_SVD_REFINE
            const float y = __builtin_amdgcn_rsqf(x);                        // (x >= 1 at every call site: no 0 * inf -- but x = inf is inf * 0)
            const float r = x * y;
            const float v = __builtin_fmaf(__builtin_fmaf(-r, r, x), 0.5f * y, r);    // r <- r + (x - r^2) / (2 r)
#else
            const float v = __builtin_amdgcn_sqrtf(x);
#endif
            return v;
        }
#endif
        return sqrtf(x);
    }
};

struct Rot2 {
    float c, s;
    __host__ __device__ bool identity() const { return c == 1.f && s == 0.f; }
};

// rows p,q of m:  (x,y) <- (c x + s y, -s x + c y)
__host__ __device__ inline void rotate_rows(Mat3& m, int p, int q, Rot2 j)
{
    if (j.identity()) return;
    for (int k = 0; k < 3; k++) {
        const float x = m.a[p][k], y = m.a[q][k];
        m.a[p][k] = j.c * x + j.s * y;
        m.a[q][k] = -j.s * x + j.c * y;
    }
}

// columns p,q of m with the same 2x2 action on (x,y)
__host__ __device__ inline void rotate_cols(Mat3& m, int p, int q, Rot2 j)
{
    if (j.identity()) return;
    for (int k = 0; k < 3; k++) {
        const float x = m.a[k][p], y = m.a[k][q];
        m.a[k][p] = j.c * x + j.s * y;
        m.a[k][q] = -j.s * x + j.c * y;
    }
}

// Jacobi rotation diagonalising the symmetric 2x2 [[x, y], [y, z]]
template <bool FAST = false>
__host__ __device__ inline Rot2 symmetric_jacobi(float x, float y, float z)
{
    using M = SvdMath<FAST>;
    const float deno = 2.f * fabsf(y);
    if (deno < FLT_MIN) return Rot2{1.f, 0.f};
    const float tau = M::div(x - z, deno);
    const float w = M::sqrt(tau * tau + 1.f);
    const float t = M::div(1.f, (tau > 0.f) ? tau + w : tau - w);
    const float sign_t = t > 0.f ? 1.f : -1.f;
    const float n = M::rsqrt(t * t + 1.f);
    // (y / |y| of Eigen's makeJacobi: +-1 exactly -- |y| >= FLT_MIN / 2 here -- so a sign copy, not a division)
    return Rot2{n, -sign_t * copysignf(1.f, y) * fabsf(t) * n};
}

struct Svd3 {
    Mat3 U, V;
    float S[3];
};

template <bool FAST = false>
__host__ __device__ inline Svd3 svd3(const Mat3& A)
{
    using M = SvdMath<FAST>;
    const float precision = 2.f * FLT_EPSILON;
    float scale = 0.f;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) scale = fmaxf(scale, fabsf(A.a[r][c]));
    if (scale == 0.f) scale = 1.f;

    Mat3 W;
    Svd3 out;
    // (FAST: one reciprocal of the scale; the singular values are scaled back by `scale` itself below, and R does not depend on it)
    const float inv_scale = FAST ? M::div(1.f, scale) : 0.f;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            W.a[r][c] = FAST ? A.a[r][c] * inv_scale : A.a[r][c] / scale;
            out.U.a[r][c] = out.V.a[r][c] = (r == c) ? 1.f : 0.f;
        }
    float max_diag = fmaxf(fabsf(W.a[0][0]), fmaxf(fabsf(W.a[1][1]), fabsf(W.a[2][2])));

    // A 3x3 Jacobi SVD converges in a handful of sweeps; the cap only guards against NaN input spinning forever.
    for (int sweep = 0; sweep < 64; sweep++) {
        bool rotated = false;
        for (int p = 1; p < 3; p++) {
            for (int q = 0; q < p; q++) {
                const float threshold = fmaxf(FLT_MIN, precision * max_diag);
                if (!(fabsf(W.a[p][q]) > threshold || fabsf(W.a[q][p]) > threshold)) continue;
                rotated = true;
                // 2x2 block [[pp, pq], [qp, qq]]: left rotation making it symmetric ...
                float m00 = W.a[p][p], m01 = W.a[p][q], m10 = W.a[q][p], m11 = W.a[q][q];
                Rot2 sym;
                const float tr = m00 + m11, d = m10 - m01;
                if (fabsf(d) < FLT_MIN) sym = Rot2{1.f, 0.f};
                else {
                    const float u = M::div(tr, d);
                    if (FAST) {
                        const float ih = M::rsqrt(1.f + u * u);
                        sym = Rot2{u * ih, ih};
                    } else {
                        const float h = sqrtf(1.f + u * u);
                        sym = Rot2{u / h, 1.f / h};
                    }
                }
                if (!sym.identity()) {
                    const float a0 = m00, a1 = m01, b0 = m10, b1 = m11;
                    m00 = sym.c * a0 + sym.s * b0;  m01 = sym.c * a1 + sym.s * b1;
                    m10 = -sym.s * a0 + sym.c * b0; m11 = -sym.s * a1 + sym.c * b1;
                }
                // ... then the Jacobi rotation of the symmetric block; left = sym * right^T
                const Rot2 right = symmetric_jacobi<FAST>(m00, m01, m11);
                const Rot2 left{sym.c * right.c + sym.s * right.s, sym.s * right.c - sym.c * right.s};
                const Rot2 right_t{right.c, -right.s};
                rotate_rows(W, p, q, left);
                rotate_cols(out.U, p, q, left);
                rotate_cols(W, p, q, right_t);
                rotate_cols(out.V, p, q, right_t);
                max_diag = fmaxf(max_diag, fmaxf(fabsf(W.a[p][p]), fabsf(W.a[q][q])));
            }
        }
        if (!rotated) break;
    }

    for (int i = 0; i < 3; i++) {
        const float d = W.a[i][i];
        out.S[i] = fabsf(d) * scale;
        if (d < 0.f)
            for (int r = 0; r < 3; r++) out.U.a[r][i] = -out.U.a[r][i];
    }
    // Eigen's selection sort, descending, swapping the matching columns of U and V -- for i = 0, 1: the FIRST maximum of S[i..2]
    // goes to i (a maximum of 0 ends the sort, which changes nothing: then the tail is all zeros and the first maximum is i itself).
    // Written with fixed indices and selects: a position computed at run time would put U, V and S into scratch memory (it did:
    // 26 scratch stores and their reloads on the solve kernel's one-lane critical path).
    auto swap_cols = [&](int i, int j, bool doit) {
        const float si = out.S[i], sj = out.S[j];
        out.S[i] = doit ? sj : si; out.S[j] = doit ? si : sj;
        for (int r = 0; r < 3; r++) {
            const float ui = out.U.a[r][i], uj = out.U.a[r][j], vi = out.V.a[r][i], vj = out.V.a[r][j];
            out.U.a[r][i] = doit ? uj : ui; out.U.a[r][j] = doit ? ui : uj;
            out.V.a[r][i] = doit ? vj : vi; out.V.a[r][j] = doit ? vi : vj;
        }
    };
    {
        const bool p1 = out.S[1] > out.S[0];
        const bool p2 = out.S[2] > (p1 ? out.S[1] : out.S[0]);
        swap_cols(0, 1, p1 && !p2);
        swap_cols(0, 2, p2);
        swap_cols(1, 2, out.S[2] > out.S[1]);
    }
    return out;
}

__host__ __device__ inline Mat3 mul_abt(const Mat3& A, const Mat3& B)   // A * B^T
{
    Mat3 C;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) C.a[r][c] = (A.a[r][0] * B.a[c][0] + A.a[r][1] * B.a[c][1]) + A.a[r][2] * B.a[c][2];
    return C;
}

__host__ __device__ inline float det3(const Mat3& M)
{
    return M.a[0][0] * (M.a[1][1] * M.a[2][2] - M.a[1][2] * M.a[2][1]) -
           M.a[0][1] * (M.a[1][0] * M.a[2][2] - M.a[1][2] * M.a[2][0]) +
           M.a[0][2] * (M.a[1][0] * M.a[2][1] - M.a[1][1] * M.a[2][0]);
}

// Kabsch rotation from the cross-covariance H = sum (a - ca)(b - cb)^T (a = fixed/after, b = moving/before):
// R = U diag(1, 1, det(U V^T)) V^T.  Also returns det and the singular values (the CPD M-step needs both).
struct Kabsch3 {
    Mat3 R;
    float S[3];
    float det;
};

// Where the fast forms break down: a quotient or a square that OVERFLOWS (a planar moving cloud gives a cross-covariance with a zero
// column; a 2 x 2 block of the sweep is then symmetrised to rounding residue and tau = (x - z) / 2|y| leaves the fp32 range).  IEEE
// arithmetic carries the infinity through (1 / inf = 0: the identity rotation); the residual steps of the fast forms turn it into
// inf - inf.  Every such case ends in NaN, and NaN spreads to U and V -- so ONE test at the end, and the decomposition again in IEEE
// arithmetic, out of line (guards inside the forms cost the solve kernel's one-lane chain 1.2 us; 7 of 400 random small problems need this).
#if defined(__HIP_DEVICE_COMPILE__)
static __attribute__((noinline, unused)) __device__ Svd3 svd3_ieee_out_of_line(const Mat3& A) { return svd3<false>(A); }
#endif

// force_ieee (device code, FAST only): the IEEE decomposition whatever the fast one would have given -- the developer switch
// MISLAM_SVD_IEEE=1 of a context (round 5: an A/B of the two on one device, tests/test_gpu_icp.py, and the way a soak's difference is
// attributed to K3 or not).  The NaN test covers S as well: the CPD M-step reads the singular values.
template <bool FAST = false>
__host__ __device__ inline Kabsch3 kabsch_rotation(const Mat3& H, bool force_ieee = false)
{
#if defined(__HIP_DEVICE_COMPILE__)
    Svd3 s;
    if (FAST && force_ieee) s = svd3_ieee_out_of_line(H);
    else {
        s = svd3<FAST>(H);
        if (FAST) {
            float chk = (fabsf(s.S[0]) + fabsf(s.S[1])) + fabsf(s.S[2]);
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) chk += fabsf(s.U.a[r][c]) + fabsf(s.V.a[r][c]);
            // Round 5 (tools/soak_rootcause.py): a RANK-DEFICIENT cross-covariance (clusters against a plane, a planar or
            // collinear moving cloud: smallest singular value ~ 0).  The directions of the null space -- and with them the sign det(U V^T) that
            // decides between a rotation and its mirror image in that direction -- are then whatever the LAST BITS of the sweep make them:
            // Eigen's arithmetic (the oracle, cpu-slam) lands on one branch, reproducibly (the oracle against itself with the points reordered:
            // 3e-6), the refined hardware forms on another (soak seed 1 case 174: singular values 1990 / 0.106 / 0, |d(R|t)|_F = 1.81 after three
            // iterations; with the IEEE forms 4e-6).  The fast forms differ from IEEE by rounding, which is harmless exactly as long as R
            // depends continuously on H: so when the decomposition says it does not -- S[2] below 1e-3 of S[0] -- the IEEE one decides.
            const bool ill = s.S[2] < 1e-3f * s.S[0];
            if (__builtin_expect(!(chk <= FLT_MAX) || ill, 0)) s = svd3_ieee_out_of_line(H);
        }
    }
#else
    (void)force_ieee;
    Svd3 s = svd3<FAST>(H);
#endif
    Kabsch3 k;
    k.det = det3(mul_abt(s.U, s.V));
    Mat3 Ud = s.U;
    for (int r = 0; r < 3; r++) Ud.a[r][2] = s.U.a[r][2] * k.det;
    k.R = mul_abt(Ud, s.V);
    for (int i = 0; i < 3; i++) k.S[i] = s.S[i];
    return k;
}

}  // namespace mislam
