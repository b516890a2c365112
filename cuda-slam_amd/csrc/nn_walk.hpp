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

// The hierarchy walk: ONE walk per wave for the (up to 64) points of its active lanes, nearer child first, each lane starting
// from its own candidate (best, bidx) -- a REAL candidate's key or (inf, 0) -- and ending with the lexicographic minimum over it
// and every point of the hierarchy.  Node, level and trail are wave-uniform, so a node's record and a leaf's points arrive
// through SCALAR loads and feed the VALU as SGPR operands (like the every-pair kernel), control flow is uniform, and all active
// lanes are busy on every box and every point.  Pending subtrees: WHICH ones is a 32-bit trail (bit l set = the sibling of the
// wave's level-l ancestor is still to be visited; the heap numbering makes it computable), their per-lane BOUNDS sit in LDS, one
// word per level and lane: st_lb[level * stride + slot].  The winner is tracked by its sorted SLOT; the global index is fetched
// once at the end -- and on an exact tie, where the lower GLOBAL index must win (rare: duplicates, or the starting candidate
// met again).  What keeps it tight is that pruning stays PER LANE: every lane computes its own bound of
// each child box against its own best, and a subtree is entered iff at least one lane needs it (a ballot), its per-lane bounds
// parked in LDS for the re-check at pop time.  The wave therefore visits the UNION of its lanes' walks -- for 64 Morton
// neighbours little more than one lane's walk -- and a wave that straddles a jump of the Z-curve pays for two compact groups, not
// for the box around both (round 1's wave-cooperative form pruned against the group's box and worst best: 77 ms).  Lanes that do
// not need a visited leaf test its points anyway: harmless for a lexicographic minimum.  Exactness is the per-lane rule
// unchanged: a lane's true neighbour lies in a subtree that lane needs at every level, so the wave enters it.
// Call with any subset of a wave's lanes active; `slot` is the lane's column in st_lb.
template <bool FMA, bool STATS>
__device__ __forceinline__ void tree_walk_wave(const NnTreeView& t, const float p[3], float& best, unsigned int& bidx,
                                               float* __restrict__ st_lb, int stride, int slot, unsigned int& n_nodes, unsigned int& n_leaves)
{
    const float4* __restrict__ pairs = t.pairs;
    const float4* __restrict__ leaf_soa = t.leaf_soa;
    const int* __restrict__ leaf_idx = t.leaf_idx;
    const int first_leaf = t.n_pad - 1;
    const float inf = __builtin_inff();
    int bslot = -1;                                            // >= 0: the winner's sorted slot; < 0: bidx is the winner
    unsigned int trail = 0;                                    // wave-uniform from here on
    int node = 0, level = 0;
    const float root_lb = box_bound<FMA>(t.boxes[0], t.boxes[1], p);
    bool have = __builtin_amdgcn_ballot_w64(root_lb <= best && root_lb < inf) != 0ull;
    auto pop = [&]() {
        have = false;
        while (trail != 0) {
            const int b = 31 - __builtin_clz(trail);              // deepest pending level
            trail &= ~(1u << b);
            const int anc = ((node + 1) >> (level - b)) - 1;      // the wave's ancestor at level b ...
            node = ((anc + 1) ^ 1) - 1;                           // ... its sibling is the pending subtree
            level = b;
            if (__builtin_amdgcn_ballot_w64(st_lb[b * stride + slot] <= best) != 0ull) { have = true; break; }
        }
    };
    auto offer = [&](float d, int s) {                          // s is wave-uniform
        const bool tie = d == best;
        const bool lt = d < best;
        best = lt ? d : best;
        bslot = lt ? s : bslot;
        if (tie) {
            const unsigned int j = (unsigned int)leaf_idx[s];
            const unsigned int jb = bslot >= 0 ? (unsigned int)leaf_idx[bslot] : bidx;
            if (j < jb) bslot = s;
        }
    };
    while (have) {
        while (have && node < first_leaf) {
            const int un = __builtin_amdgcn_readfirstlane(node);   // uniform by construction; say so, so the loads below are scalar
            if (STATS) n_nodes += 1;
            const float4* __restrict__ rec = pairs + 3 * (size_t)un;
            const float4 a = rec[0], b = rec[1], c = rec[2];
            const float lbl = box_bound<FMA>(make_float4(a.x, a.y, a.z, 0.f), make_float4(a.w, b.x, b.y, 0.f), p);
            const float lbr = box_bound<FMA>(make_float4(b.z, b.w, c.x, 0.f), make_float4(c.y, c.z, c.w, 0.f), p);
            const unsigned long long ml = __builtin_amdgcn_ballot_w64(lbl <= best && lbl < inf);
            const unsigned long long mr = __builtin_amdgcn_ballot_w64(lbr <= best && lbr < inf);
            if ((ml | mr) == 0ull) { pop(); continue; }
            bool left_first = mr == 0ull;
            if (ml != 0ull && mr != 0ull)                          // both needed: the child nearer to most lanes first
                left_first = 2 * __builtin_popcountll(__builtin_amdgcn_ballot_w64(lbl <= lbr)) >= __builtin_popcountll(__builtin_amdgcn_ballot_w64(true));
            node = 2 * un + (left_first ? 1 : 2);
            level += 1;
            if (ml != 0ull && mr != 0ull) {
                trail |= 1u << level;
                st_lb[level * stride + slot] = left_first ? lbr : lbl;   // this lane's bound of the subtree left pending
            }
        }
        if (have) {
            const int leaf = __builtin_amdgcn_readfirstlane(node) - first_leaf;
            if (STATS) n_leaves += 1;
            const int slot0 = leaf * TREE_LEAF;
            const float4* __restrict__ lp = leaf_soa + (size_t)leaf * (3 * TREE_LEAF / 4);
#pragma unroll
            for (int c4 = 0; c4 < TREE_LEAF / 4; c4++) {
                const float4 X = lp[c4], Y = lp[TREE_LEAF / 4 + c4], Z = lp[2 * (TREE_LEAF / 4) + c4];
                offer(sq3<FMA>(X.x - p[0], Y.x - p[1], Z.x - p[2]), slot0 + 4 * c4);
                offer(sq3<FMA>(X.y - p[0], Y.y - p[1], Z.y - p[2]), slot0 + 4 * c4 + 1);
                offer(sq3<FMA>(X.z - p[0], Y.z - p[1], Z.z - p[2]), slot0 + 4 * c4 + 2);
                offer(sq3<FMA>(X.w - p[0], Y.w - p[1], Z.w - p[2]), slot0 + 4 * c4 + 3);
            }
            pop();
        }
    }
    if (bslot >= 0) bidx = (unsigned int)leaf_idx[bslot];
}

}  // namespace mislam
