// Device-side pieces shared by the two exact searches (nn_tree.hip, nn_grid.hip): the distance in the reference's operation
// order, the rounded box bound, the hierarchy walk and the XCD-aware chunk mapping.
//
// Contract of every search built from these (FindCorrespondences, cudacommon.cu:57-77 / common.cpp:446-462): idx[i] = argmin_j
// |after[j] - before[i]|^2 under strict '<' with the lowest index winning ties, the distance evaluated with the same fp32
// operation sequence.  Exactness rests on two rules:
//   * a candidate is accepted iff (d, j) is lexicographically smaller than the running (best, bidx) -- the order in which
//     candidates are met does not matter for a lexicographic minimum;
//   * a set of candidates is skipped only if a lower bound lb of their distances is STRICTLY greater than the running best, and
//     lb is computed so that rounding cannot lift it above any skipped candidate's rounded distance.  For a box [lo,hi] the bound
//     uses the very operations of a distance: per axis q - s >= lo - s and s - q >= s - hi in real arithmetic, rounding is
//     monotonic, so |fl(q - s)| >= fl(gap) with gap = max(lo - s, s - hi, 0), and squaring / summing in the distance's own order
//     keeps the inequality.  A skipped point can neither win nor tie.
#pragma once
#include <hip/hip_runtime.h>

#include "nn_tree.h"

namespace mislam {

// A pointer into memory this kernel never writes, in the CONSTANT address space: a wave-uniform load through it is a scalar load whatever the compiler can
// or cannot prove about the kernel's own stores and atomics.  (Round 6: a persistent form of the search that draws its chunks off a global counter was
// built and measured -- ONE atomic in the kernel made the compiler give up the scalar form of every uniform load behind it, hierarchy boxes, leaf points,
// the state: 38 of 55; through this address space they stay scalar by construction.  The persistent form itself lost 30 % and is gone, DESIGN section 4
// K1g; the arrays in question are read-only for the launch, which is all the constant address space asserts, so the walk keeps it.)
template <class T> using cptr = const T __attribute__((address_space(4)))*;
template <class T> __device__ __forceinline__ cptr<T> as_constant(const T* p) { return (cptr<T>)(unsigned long long)p; }
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <bool FMA>
__device__ __forceinline__ float sq3(float dx, float dy, float dz)
{
    if constexpr (FMA) return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    else return (dx * dx + dy * dy) + dz * dz;
}

// lower bound of |q - s|^2 over q in [lo,hi], rounded like a distance
template <bool FMA>
__device__ __forceinline__ float box_bound(const float4 lo, const float4 hi, const float s[3])
{
    const float ex = fmaxf(fmaxf(lo.x - s[0], s[0] - hi.x), 0.f);
    const float ey = fmaxf(fmaxf(lo.y - s[1], s[1] - hi.y), 0.f);
    const float ez = fmaxf(fmaxf(lo.z - s[2], s[2] - hi.z), 0.f);
    return sq3<FMA>(ex, ey, ez);
}

// Which chunk of the Morton-sorted moving cloud workgroup `b` of `grid` takes: runs of `run` consecutive chunks per XCD, the 8
// XCDs taking neighbouring runs (workgroups are dealt to the XCDs round-robin, so block b runs with blocks b + 8k).  A
// permutation of [0, grid): speed only.
__device__ __forceinline__ unsigned int xcd_chunk(unsigned int b, unsigned int grid, unsigned int run)
{
    if (run <= 1) return b;
    const unsigned int whole = grid / (8u * run) * (8u * run);
    if (b >= whole) return b;
    const unsigned int x = b & 7u, j = b >> 3;
    return ((j / run) * 8u + x) * run + j % run;
}

__device__ __forceinline__ void unpack_start(unsigned long long k0, float& best, unsigned int& bidx)
{
    // starting candidate: the key already posted (KEY_INIT -> none).  (inf, 0): nothing at +inf is ever accepted
    const unsigned int hi0 = (unsigned int)(k0 >> 32);
    best = hi0 < 0x7f800000u ? __uint_as_float(hi0) : __builtin_inff();
    bidx = hi0 < 0x7f800000u ? (unsigned int)(k0 & 0xffffffffull) : 0u;
}

// The hierarchy walk: ONE walk per wave for the (up to 64) points of its active lanes, each lane starting from its own candidate
// (best, bidx) -- a REAL candidate's key or (inf, 0) -- and ending with the lexicographic minimum over it and every point of the
// hierarchy.  Node and level are wave-uniform, so boxes and leaf points arrive through SCALAR loads and feed the VALU as SGPR
// operands (like the every-pair kernel), control flow is uniform, and all active lanes are busy on every box and every point.
// What keeps it tight is that pruning stays PER LANE: every lane computes its own bound of each child box against its own best,
// and a subtree is entered iff at least one lane needs it (a ballot).  The wave therefore visits the UNION of its lanes' walks --
// for 64 Morton neighbours a few times one lane's walk -- and a wave that straddles a jump of the Z-curve pays for two compact
// groups, not for the box around both (round 1's wave-cooperative form pruned against the group's box and worst best: 77 ms
// per search).  Lanes that do not need a visited leaf test its points anyway: harmless for a lexicographic minimum.
// EIGHT children per step: the implicit heap keeps the 2^k descendants k levels below a node CONTIGUOUS (nodes ((n + 1) << k) - 1
// ...), so one step fetches eight boxes (192 bytes, three s_load_dwordx16), tests them all per lane and goes three levels down.
// (A binary step per level -- 48-byte sibling records, per-lane bounds parked in LDS -- measured the same time with three times
// the dependent loads: the walk is bound by vector-instruction issue, ~180 per wide step and ~150 per leaf, not by latency.)
// State: node and wide level (wave-uniform) and, per wide level, an 8-bit mask of the children still to visit (one 64-bit
// scalar: no LDS at all).  The winner is tracked by its sorted SLOT; the global index is fetched once at the end -- and on an
// exact tie, where the lower GLOBAL index must win (rare: duplicates, or the starting candidate met again).  Children that are
// leaves are scanned inside the step that tested their boxes, each re-checked against the bests as they stand then; interior children wait in the mask and are entered without a re-check (their own step
// prunes).  First the child that is nearest for most lanes, then the others in index order.  Exactness is the per-lane rule
// unchanged: a lane's true neighbour lies in a child that lane needs at every step, needed children are never dropped, skipped
// ones have a bound strictly above that lane's best.  Call with any subset of a wave's lanes active.
// b: the node's lo.x in NnTreeView::boxes6 (siblings interleaved: its other components follow two floats apart)
template <bool FMA, class P>
__device__ __forceinline__ float box6_bound(P b, const float s[3])
{
    const float ex = fmaxf(fmaxf(b[0] - s[0], s[0] - b[6]), 0.f);
    const float ey = fmaxf(fmaxf(b[2] - s[1], s[1] - b[8]), 0.f);
    const float ez = fmaxf(fmaxf(b[4] - s[2], s[2] - b[10]), 0.f);
    return sq3<FMA>(ex, ey, ez);
}

// the bounds of the two boxes of one PAIR (twelve floats: lo.x lo.x' lo.y lo.y' lo.z lo.z' hi.x hi.x' ...) at once: the gaps, the squares
// and the sums as packed fp32 operations (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 -- the same IEEE operations as the single ones, two
// results per issue slot), the pair's components coming straight out of the scalar loads as register pairs
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool FMA, class P>
__device__ __forceinline__ f32x2 box6_bound2(P pr, const float s[3])
{
    const f32x2 sx = {s[0], s[0]}, sy = {s[1], s[1]}, sz = {s[2], s[2]};
    const f32x2 ax = (f32x2){pr[0], pr[1]} - sx, bx = sx - (f32x2){pr[6], pr[7]};
    const f32x2 ay = (f32x2){pr[2], pr[3]} - sy, by = sy - (f32x2){pr[8], pr[9]};
    const f32x2 az = (f32x2){pr[4], pr[5]} - sz, bz = sz - (f32x2){pr[10], pr[11]};
    f32x2 ex, ey, ez;
    ex.x = fmaxf(fmaxf(ax.x, bx.x), 0.f); ex.y = fmaxf(fmaxf(ax.y, bx.y), 0.f);
    ey.x = fmaxf(fmaxf(ay.x, by.x), 0.f); ey.y = fmaxf(fmaxf(ay.y, by.y), 0.f);
    ez.x = fmaxf(fmaxf(az.x, bz.x), 0.f); ez.y = fmaxf(fmaxf(az.y, bz.y), 0.f);
    if constexpr (FMA) return __builtin_elementwise_fma(ez, ez, __builtin_elementwise_fma(ey, ey, ex * ex));
    else return (ex * ex + ey * ey) + ez * ez;
}

template <bool FMA, bool STATS>
__device__ __forceinline__ void tree_walk_wide(const NnTreeView& t, const float p[3], float& best, unsigned int& bidx,
                                               unsigned int& n_nodes, unsigned int& n_leaves, bool nearest_first = false,
                                               unsigned int* walk_counts = nullptr     // (STATS) [0] leaves with a candidate at or below some lane's best, [1] leaf children looked at, [2] votes for the nearest child, [3] pops, [4] leaves whose sequential offers ran
                                               )
{
    const cptr<float> boxes6 = as_constant(t.boxes6);
    const cptr<f32x4> leaf_soa = as_constant(reinterpret_cast<const f32x4*>(t.leaf_soa));
    const int* __restrict__ leaf_idx = t.leaf_idx;               // (per-lane gathers: the offers' tie rule, the winner's index at the end)
    const cptr<int> leaf_idx_c = as_constant(t.leaf_idx);        // (a leaf's eight indices at once: wave-uniform)
    const int H = t.height, first_leaf = t.n_pad - 1, real_leaves = t.n_leaves;
    const float inf = __builtin_inff();
    int bslot = -1;                                            // >= 0: the winner's sorted slot; < 0: bidx is the winner
    auto offer = [&](float d, int s) {                          // s is wave-uniform
        const bool tie = d == best;
        const bool lt = d < best;
        best = lt ? d : best;
        bslot = lt ? s : bslot;
        if (tie) {
            const unsigned int j = (unsigned int)leaf_idx_c[s];   // (s is wave-uniform: a scalar load)
            const unsigned int jb = bslot >= 0 ? (unsigned int)leaf_idx[bslot] : bidx;
            if (j < jb) bslot = s;
        }
    };
    // A leaf's distances first (packed fp32: the same IEEE operations, two per issue slot), then ONE question to the wave: does any
    // lane see a point at or below its best?  Mostly not -- a walk that starts from real candidates is mainly there to prove that
    // nothing closer exists -- and then the leaf is done; otherwise the sequential offers, which settle ties by global index.
    auto scan_leaf = [&](int leaf) {                            // leaf is wave-uniform: its points arrive through scalar loads
        if (STATS) n_leaves += 1;
        const int slot0 = leaf * TREE_LEAF;
        const cptr<f32x4> lp = leaf_soa + (size_t)leaf * (3 * TREE_LEAF / 4);
        f32x2 d[TREE_LEAF / 2];
        float m = inf;
#pragma unroll
        for (int c4 = 0; c4 < TREE_LEAF / 4; c4++) {
            const f32x4 X = lp[c4], Y = lp[TREE_LEAF / 4 + c4], Z = lp[2 * (TREE_LEAF / 4) + c4];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                f32x2 dx, dy, dz;
                dx.x = (h ? X.z : X.x) - p[0]; dx.y = (h ? X.w : X.y) - p[0];
                dy.x = (h ? Y.z : Y.x) - p[1]; dy.y = (h ? Y.w : Y.y) - p[1];
                dz.x = (h ? Z.z : Z.x) - p[2]; dz.y = (h ? Z.w : Z.y) - p[2];
                f32x2 dd;
                if constexpr (FMA) dd = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                else dd = (dx * dx + dy * dy) + dz * dz;
                d[2 * c4 + h] = dd;
                m = fminf(m, fminf(dd.x, dd.y));
            }
        }
        if (__builtin_amdgcn_ballot_w64(m <= best) == 0ull) return;
        if (STATS && walk_counts) walk_counts[0] += 1u;
        // Round 6: half the leaves a warm walk scans get this far (profiles/r06_search_budget.md), and the sixteen-way select chain of the
        // sequential offers below was a fifth of a walking wave's instructions.  The offers only matter where an INDEX decides between two
        // points of THIS walk: a lane that meets its leaf minimum twice, or ties a best it found in an earlier leaf.  Everywhere else the
        // outcome of the eight offers is "the leaf's minimum, if it is lexicographically smaller than (best, bidx)" -- the one slot that holds
        // m, its global index from the leaf's eight (a scalar load: the leaf is wave-uniform), whatever the intermediate steps did.  The
        // commonest tie by far is a lane meeting its own STARTING candidate again (the previous match is a point of this tree): equal
        // index, nothing changes.
        {
            int km = 0;
            unsigned int cnt = 0u;
#pragma unroll
            for (int k = TREE_LEAF - 1; k >= 0; k--) {
                const float dk = (k & 1) ? d[k >> 1].y : d[k >> 1].x;
                const bool e = dk == m;
                km = e ? k : km;
                cnt += e ? 1u : 0u;
            }
            const bool tie = m == best;
            if (__builtin_amdgcn_ballot_w64(cnt > 1u || (tie && bslot >= 0)) == 0ull) {
                bool take = m < best;
                if (__builtin_amdgcn_ballot_w64(tie) != 0ull) {
                    static_assert(TREE_LEAF == 8, "a leaf's indices are two 16-byte scalar loads");
                    const cptr<i32x4> ip = (cptr<i32x4>)(leaf_idx_c + slot0);
                    const i32x4 I0 = ip[0], I1 = ip[1];
                    const int idx8[8] = {I0.x, I0.y, I0.z, I0.w, I1.x, I1.y, I1.z, I1.w};
                    int j = idx8[0];
#pragma unroll
                    for (int k = 1; k < TREE_LEAF; k++) j = km == k ? idx8[k] : j;
                    take = take || (tie && (unsigned int)j < bidx);      // (bslot < 0 here: bidx is the running winner's index)
                }
                best = take ? m : best;
                bslot = take ? slot0 + km : bslot;
                return;
            }
        }
        if (STATS && walk_counts) walk_counts[4] += 1u;
#pragma unroll
        for (int k = 0; k < TREE_LEAF / 2; k++) {
            offer(d[k].x, slot0 + 2 * k);
            offer(d[k].y, slot0 + 2 * k + 1);
        }
    };
    {
        const float root_lb = box6_bound<FMA, cptr<float>>(boxes6 + tree_box_offset(0), p);
        if (__builtin_amdgcn_ballot_w64(root_lb <= best && root_lb < inf) == 0ull) return;
    }
    if (H == 0) {
        scan_leaf(0);
    } else {
        const int k0 = H % 3 == 0 ? 3 : H % 3;                  // binary levels of the root step; every later step takes three
        unsigned long long pend = 0ull;                         // byte w: children of the wave's wide-level-w ancestor still to visit
        int node = 0, wl = 0, level = 0;                        // wave-uniform
        bool have = true;
        while (have) {
            const int un = __builtin_amdgcn_readfirstlane(node);   // uniform by construction; say so, so the loads below are scalar
            const int k = wl == 0 ? k0 : 3;
            const int base = ((un + 1) << k) - 1;                   // first of the 2^k descendants k levels down
            const int clevel = level + k;
            if (STATS) n_nodes += 1;
            const cptr<float> bp = boxes6 + (size_t)((base + 1) >> 1) * 12;   // base is odd: the first of a pair of siblings
            float lb[8];
            unsigned int mask = 0u;
            // (a root step of fewer than three levels reads boxes past its children: masked off below)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const f32x2 b2 = box6_bound2<FMA, cptr<float>>(bp + 6 * j, p);
                lb[j] = b2.x;
                lb[j + 1] = b2.y;
                if (__builtin_amdgcn_ballot_w64(lb[j] <= best) != 0ull) mask |= 1u << j;
                if (__builtin_amdgcn_ballot_w64(lb[j + 1] <= best) != 0ull) mask |= 1u << (j + 1);
            }
            // The empty padding boxes (bound inf) would pass `lb <= best` for a lane that has no candidate yet (best = inf): they
            // are kept out by position -- level clevel's node i is empty iff its first leaf i << (H - clevel) is >= real_leaves
            {
                const int sh = H - clevel;
                const int real = ((real_leaves + (1 << sh) - 1) >> sh) - (base - ((1 << clevel) - 1));   // real nodes from `base` on
                const int kids = 1 << k;
                const int cnt = real < kids ? (real < 0 ? 0 : real) : kids;
                mask &= (1u << cnt) - 1u;
            }
            bool descend = false;
            if (clevel == H) {
                // the children are leaves: scan the needed ones now, each re-checked against the bests as they stand
                // (one copy of the scan code: a loop over the set bits, the bound picked by a select chain)
#pragma unroll 1
                for (unsigned int m = mask; m != 0u; m &= m - 1u) {
                    const int j = __builtin_ctz(m);
                    if (STATS && walk_counts) walk_counts[1] += 1u;
                    float lbj = lb[0];
#pragma unroll
                    for (int c = 1; c < 8; c++) lbj = j == c ? lb[c] : lbj;
                    if (__builtin_amdgcn_ballot_w64(lbj <= best) != 0ull) scan_leaf(base + j - first_leaf);
                }
            } else if (mask != 0u) {
                // first the child that is the nearest one for most lanes -- worth a vote only while some lane has no candidate
                // yet (a good first descent is all its pruning); lanes that came with one prune by it whatever the order
                int f = __builtin_ctz(mask);
                if (nearest_first || __builtin_amdgcn_ballot_w64(!(best < inf)) != 0ull) {
                    if (STATS && walk_counts) walk_counts[2] += 1u;
                    float minb = inf;
#pragma unroll
                    for (int j = 0; j < 8; j++) minb = ((mask >> j) & 1u) ? fminf(minb, lb[j]) : minb;
                    int fv = -1;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int v = ((mask >> j) & 1u) ? (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(lb[j] == minb)) : -1;
                        if (v > fv) { fv = v; f = j; }
                    }
                }
                const unsigned long long rest = (unsigned long long)(mask & ~(1u << f));
                pend = (pend & ~(0xffull << (8 * wl))) | (rest << (8 * wl));
                node = base + f;
                level = clevel;
                wl += 1;
                descend = true;
            }
            if (!descend) {
                // next pending child, deepest wide level first
                have = pend != 0ull;
                if (have) {
                    if (STATS && walk_counts) walk_counts[3] += 1u;
                    const int w = (63 - __builtin_clzll(pend)) >> 3;
                    const unsigned int slot = (unsigned int)(pend >> (8 * w)) & 0xffu;
                    const int j = __builtin_ctz(slot);
                    pend &= ~(1ull << (8 * w + j));
                    const int lw = w == 0 ? 0 : k0 + 3 * (w - 1);   // binary level of wide level w
                    const int parent = ((un + 1) >> (level - lw)) - 1;   // the wave's ancestor there
                    const int kw = w == 0 ? k0 : 3;
                    node = ((parent + 1) << kw) - 1 + j;
                    level = lw + kw;
                    wl = w + 1;
                }
            }
        }
    }
    if (bslot >= 0) bidx = (unsigned int)leaf_idx[bslot];
}

}  // namespace mislam
