// Gumbel search (sequential halving at the root, completed-Q policy improvement below it) with CACHED SELECTION
// DECISIONS, leaves evaluated on the matrix cores; the whole simulation loop in one launch.
//
// Under the Gumbel rules the action chosen at a node (mcts/action_selection.py:73-118 with
// mcts/qtransforms.py:25-77) is a pure function of that node's own record - priors, child visits, child values and
// rewards, raw value; at the root also the tree's Gumbel noise, legality mask and visit schedule - and a record only
// changes when the node lies on the path of a simulation (MCTS.backward, mcts/mcts.py:143-164, and the link written by
// MCTS.expand, :116-141).  So instead of scoring every node of the path while walking down (a chain of dependent
// record reads and float64 arithmetic per level, which is what bounds the level-by-level kernels), each simulation
//   1. walks down a table of decisions kept in LDS (one 4-byte read per level, mcts/mcts.py:98-114),
//   2. evaluates the leaf on the matrix cores (network.py:105-126),
//   3. expands and backs up (tree_ops.h expand_backup), and
//   4. re-decides exactly the nodes whose records changed - the path and the new leaf - all at once:
//      one LANE per node, looping over the actions in registers (no cross-lane reductions, no idle action lanes),
//      the path nodes of the block's whole tile on one work list (made before the network runs) dealt evenly over
//      its threads, the new leaf decided from its logits where they have a clear leader.
// A decision is an argmax: it is taken from cheap scores when their lead is certified to exceed what the cheap and the
// reference evaluation can differ by (gumbel_decide).  Everything else - and every other case - uses the expressions
// and operand order of the lane-group kernels (tree_ops.h select_at and what it calls), with the group reductions
// replaced by their serial equivalents, so trees come out bit for bit the same.
// PUCT does not qualify: its score contains a fresh tie-break draw per selection call (mcts/action_selection.py:46).
#pragma once
#include <utility>
#include "net_mfma.h"

namespace mz {

// Record layout of a Gumbel tree with A actions and visit / index rows of IW bytes per entry (2 or 4), as mz_create
// lays it out.
template <int A, int IW>
struct GumbelLayout {
    static_assert(A >= 2 && A <= 16 && (IW == 2 || IW == 4), "one 16-lane group per tree; int16 or int32 rows");
    static constexpr int ru(int x, int m) { return (x + m - 1) / m * m; }
    static constexpr int off_cval = 32;
    static constexpr int off_crew = off_cval + 8 * A;
    static constexpr int off_prior = off_crew + 8 * A;
    static constexpr int off_cvis = off_prior + ru(4 * A, 8);
    static constexpr int off_cidx = off_cvis + ru(IW * A, 4);
    static constexpr int rec = ru(off_cidx + IW * A, 64);
};

// np.sum over A values in NumPy's order (numpy/_core/src/umath/loops_utils.h.src pairwise_sum below its block size):
// sequential below 8 elements, else 8 strided accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and the
// tail added in order.  The serial twin of gsum_np (device_common.h).
template <int A, typename T>
__device__ __forceinline__ T np_sum_serial(const T (&x)[A])
{
    if constexpr (A < 8) {
        T r = x[0];
#pragma unroll
        for (int i = 1; i < A; i++) r += x[i];
        return r;
    } else {
        constexpr int lim = A - A % 8;
        T r[8];
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = x[j];
#pragma unroll
        for (int i = 8; i < lim; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] += x[i + j];
        }
        T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
        for (int i = lim; i < A; i++) res += x[i];
        return res;
    }
}

// Field extraction from a record held as 16-byte chunks in registers (compile-time loops: static_for, device_common.h).
template <int K, int NCH>
__device__ __forceinline__ uint32_t rec_word(const uint4 (&ch)[NCH])
{
    const uint4 &v = ch[K / 4];
    if constexpr (K % 4 == 0) return v.x;
    else if constexpr (K % 4 == 1) return v.y;
    else if constexpr (K % 4 == 2) return v.z;
    else return v.w;
}
template <int OFF, int NCH>
__device__ __forceinline__ double rec_f64(const uint4 (&ch)[NCH])
{
    return __builtin_bit_cast(double, ((unsigned long long)rec_word<OFF / 4 + 1>(ch) << 32) | (unsigned long long)rec_word<OFF / 4>(ch));
}
// an entry of the visit / index rows: int16 (IW = 2) or int32 (IW = 4) at byte OFF
template <int OFF, int IW, int NCH>
__device__ __forceinline__ int rec_int(const uint4 (&ch)[NCH])
{
    if constexpr (IW == 2) return (int)(int16_t)(rec_word<OFF / 4>(ch) >> ((OFF & 3) * 8));
    else return (int)rec_word<OFF / 4>(ch);
}

// A decision: the action the selection rule picks at a node and where it leads.
//   word = (children_index[node, action] + 1) << 8 | action      (an unexpanded edge gives 0 in the upper bits)
// One lane evaluates one node from its record in global memory.  `gum`: the tree's Gumbel noise [A] (root only),
// `cv_row`: the tree's sequential-halving schedule (root only).
// The record as 16-byte chunks in registers (it is 64-byte aligned); every field sits at a compile-time position.
template <int A, int IW>
struct GumbelRec {
    static constexpr int NCH = (GumbelLayout<A, IW>::off_cidx + IW * A + 15) / 16;
    uint4 ch[NCH];
};

// Loads the chunks of `mask` (bit k = bytes 16 k .. 16 k + 15 of the record).
template <int A, int IW>
__device__ __forceinline__ void rec_load(GumbelRec<A, IW> &r, const char *rec, unsigned mask)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(rec);
    static_for<GumbelRec<A, IW>::NCH>([&](auto K) {
        constexpr int k = decltype(K)::value;
        if ((mask >> k) & 1u) r.ch[k] = src[k];
    });
}

template <int A, int IW>
__device__ __forceinline__ int gumbel_decide(const Dev &d, const GumbelRec<A, IW> &rec, bool is_root, unsigned long long legal,
                                             const double *gum, const int32_t *cv_row, long long *ld_cycles = nullptr)
{
    using GL = GumbelLayout<A, IW>;
    constexpr int NCH = GumbelRec<A, IW>::NCH;
    const uint4 (&ch)[NCH] = rec.ch;
#ifdef MZ_STAMPS
    const long long ld_t0 = clock64();
    if (ld_cycles) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); *ld_cycles += clock64() - ld_t0; }
#endif
    const double inf = __builtin_huge_val();
    const double raw = rec_f64<24>(ch);
    double q[A];
    float pf[A];
    int cv[A], ci[A];
    int vsum = 0, vmax = -2147483647;
    static_for<A>([&](auto I) {
        constexpr int a = decltype(I)::value;
        const double v = rec_f64<GL::off_cval + 8 * a>(ch), r = rec_f64<GL::off_crew + 8 * a>(ch);
        q[a] = r + d.discount * v;                                     // Tree.qvalues, mcts/tree.py:55-58
        pf[a] = __builtin_bit_cast(float, rec_word<(GL::off_prior + 4 * a) / 4>(ch));
        cv[a] = rec_int<GL::off_cvis + IW * a, IW>(ch);
        ci[a] = rec_int<GL::off_cidx + IW * a, IW>(ch);
        vsum += cv[a];
        vmax = cv[a] > vmax ? cv[a] : vmax;
    });
    // compute_mixed_value, mcts/qtransforms.py:51-77: only its entries at unvisited actions are used (:41), and there
    // the reference divides the weighted q by 1.0
    float e[A];
#pragma unroll
    for (int a = 0; a < A; a++) e[a] = np_expf(pf[a]);
    const float es = np_sum_serial<A, float>(e);
    const float tiny = 1.17549435e-38f;
    double pq[A];
#pragma unroll
    for (int a = 0; a < A; a++) {
        float p = __fdiv_rn(e[a], es);
        p = p < tiny ? tiny : p;                                       // np.maximum(tiny, p): a NaN stays a NaN
        pq[a] = cv[a] > 0 ? (double)p * q[a] : 0.0;
    }
    const double sum_pq = np_sum_serial<A, double>(pq);
    const double wq = sum_pq / (double)1.0f;
    const double mixed = (raw + (double)vsum * wq) / (double)(vsum + 1);
    // qtransform_completed_by_mix_value, mcts/qtransforms.py:25-48
    double comp[A];
#pragma unroll
    for (int a = 0; a < A; a++) comp[a] = cv[a] > 0 ? q[a] : mixed;
    double lo = comp[0], hi = comp[0];
#pragma unroll
    for (int a = 1; a < A; a++) {
        const double y = comp[a];
        lo = (y < lo || y != y) ? y : lo;                              // np.min / np.max: a NaN wins
        hi = (y > hi || y != y) ? y : hi;
    }
    double den = hi - lo;
    den = den < 1e-8 ? 1e-8 : den;                                     // np.maximum: a NaN stays a NaN
    const double scale = (d.c_visit + (double)vmax) * d.c_scale;
    // The chosen action needs an argmax, not the scores themselves.  The scores are first evaluated cheaply - one
    // reciprocal per denominator instead of a correctly rounded division per action, float32 exponentials of
    // max-shifted arguments - and the cheap argmax is taken when the best score leads the runner-up by more than the two
    // evaluations can differ; every other case (near ties, exact ties, NaN or infinite scores, arguments near the range
    // of exp) is evaluated with the reference's operations below.  Bounds, with t = (comp - lo) / den in [0, 1]:
    //  * rden = 1/den after two Newton steps is within 3e-16 (relative) of the quotient's factor, so
    //    |scale * t~ - scale * t| <= 1e-15 * scale, plus one rounding of each sum that follows (2.3e-16 relative);
    //  * root: the same additions in the same order, so scores differ by at most 1e-15 * scale + a few ulp of the
    //    operands; the lead demanded is 1e-10 * (scale + |best| + 1);
    //  * interior: softmax(x)[a] is invariant under a common shift; with z = (x - xmax) * log2(e) split in float64 into
    //    an integer n and a fraction |f| <= 1/2 (z itself is within 1e-13 of its exact value for z >= -200, and terms
    //    below 2^-126 only matter as zeros), 2^f from v_exp_f32(float(f)) is within 2e-7 (relative: conversion 2e-8,
    //    one ulp 1.2e-7) and the scaling by 2^n is exact; the float32 sum of A <= 16 terms (>= 1: the largest is
    //    2^0) adds 1e-6, the reciprocal and product 2e-7, the float32 visit share 3e-7.  A cheap score is therefore
    //    within 2e-6 of the exact one; the lead demanded is 1e-5.
    int best = 0;
    bool sure = false;
#ifdef MZ_STAMPS
    if (ld_cycles) { if (scale == 12345.678 && den == 3.25) ld_cycles[1]++; ld_cycles[3] += clock64() - ld_t0; }
#endif
#ifndef MZ_DECIDE_EXACT_ONLY
    {
        double rden = __builtin_amdgcn_rcp(den);
        rden = __builtin_fma(__builtin_fma(-den, rden, 1.0), rden, rden);
        rden = __builtin_fma(__builtin_fma(-den, rden, 1.0), rden, rden);
        if (is_root) {
            const int considered = cv_row[vsum < d.L.S ? vsum : d.L.S - 1];
            float lmax = pf[0];
#pragma unroll
            for (int a = 1; a < A; a++) lmax = pf[a] > lmax ? pf[a] : lmax;
            double b1 = -inf, b2 = -inf;
            bool bad = false;
#pragma unroll
            for (int a = 0; a < A; a++) {
                const double cq = scale * ((comp[a] - lo) * rden);
                double sc = (gum[a] + (double)(pf[a] - lmax)) + cq;
                bad = bad || sc != sc;
                sc = sc < -1e9 ? -1e9 : sc;
                sc = (cv[a] == considered && ((legal >> a) & 1ull)) ? sc : -inf;
                if (sc > b1) { b2 = b1; b1 = sc; best = a; }
                else if (sc > b2) b2 = sc;
            }
            sure = !bad && b1 > -1e300 && b1 < 1e300 && (b1 - b2) > 1e-10 * (scale + __builtin_fabs(b1) + 1.0);
        } else {
            double x[A];
            double xmax = -inf;
#pragma unroll
            for (int a = 0; a < A; a++) {
                x[a] = (double)pf[a] + scale * ((comp[a] - lo) * rden);
                xmax = x[a] > xmax ? x[a] : xmax;
            }
            float ef[A], es = 0.0f;
#pragma unroll
            for (int a = 0; a < A; a++) {
                // exp(x - xmax) = 2^n * 2^f with n an integer and |f| <= 1/2, both taken in float64: the float32
                // conversion of f is then off by at most 3e-8 and v_exp_f32 by one ulp
                const double z = (x[a] - xmax) * 1.4426950408889634;
                const double n = __builtin_rint(z < -200.0 ? -200.0 : z);
                ef[a] = __builtin_ldexpf(__builtin_amdgcn_exp2f((float)(z - n)), (int)n);
                es += ef[a];
            }
            const float rs = __builtin_amdgcn_rcpf(es), rv = __builtin_amdgcn_rcpf((float)(1 + vsum));
            float b1 = -__builtin_huge_valf(), b2 = -__builtin_huge_valf();
#pragma unroll
            for (int a = 0; a < A; a++) {
                const float sc = ef[a] * rs - (float)cv[a] * rv;
                if (sc > b1) { b2 = b1; b1 = sc; best = a; }
                else if (sc > b2) b2 = sc;
            }
            // a NaN anywhere leaves the lead NaN (comparison false); exp(x) must stay finite and non-zero in float64
            sure = (b1 - b2) > 1e-5f && __builtin_fabs(xmax) < 700.0;
        }
    }
#endif
#ifdef MZ_STAMPS
    if (ld_cycles) { ld_cycles[1] += sure ? 0 : 1; ld_cycles[2] += __builtin_amdgcn_ballot_w64(!sure) ? 1 : 0; ld_cycles[4] += clock64() - ld_t0 + best * 0; }
#endif
    if (!sure) {
    double score[A];
    if (is_root) {   // gumbel_muzero_root_action_selection, mcts/action_selection.py:73-96 with gumbel_score :147-157
        const int considered = cv_row[vsum < d.L.S ? vsum : d.L.S - 1];
        float lmax = pf[0];
#pragma unroll
        for (int a = 1; a < A; a++) lmax = pf[a] > lmax ? pf[a] : lmax;
#pragma unroll
        for (int a = 0; a < A; a++) {
            const double cq = scale * ((comp[a] - lo) / den);
            const float l = pf[a] - lmax;
            double sc = (gum[a] + (double)l) + cq;
            sc = sc < -1e9 ? -1e9 : sc;                                // np.maximum(-1e9, .) keeps a NaN
            sc = sc + (cv[a] == considered ? 0.0 : -inf);
            sc = sc != sc ? inf : sc;                                  // np.argmax: a NaN is the maximum
            score[a] = ((legal >> a) & 1ull) ? sc : -inf;
        }
    } else {         // gumbel_muzero_interior_action_selection, mcts/action_selection.py:99-118
        double e2[A];
#pragma unroll
        for (int a = 0; a < A; a++) {
            const double cq = scale * ((comp[a] - lo) / den);
            e2[a] = exp((double)pf[a] + cq);
        }
        const double es2 = np_sum_serial<A, double>(e2);
#pragma unroll
        for (int a = 0; a < A; a++) {
            double sc = e2[a] / es2 - (double)cv[a] / (double)(1 + vsum);
            score[a] = sc != sc ? inf : sc;
        }
    }
    best = 0;
    double bs = score[0];
#pragma unroll
    for (int a = 1; a < A; a++) {
        if (score[a] > bs) { bs = score[a]; best = a; }                // first maximum
    }
    }
    int child = 0;
#pragma unroll
    for (int a = 0; a < A; a++) child = a == best ? ci[a] : child;
#ifdef MZ_STAMPS
    if (ld_cycles) ld_cycles[5] += clock64() - ld_t0 + child * 0;
#endif
    return ((child + 1) << 8) | best;
}

template <int TM> struct Log2 { static constexpr int value = 1 + Log2<TM / 2>::value; };
template <> struct Log2<1> { static constexpr int value = 0; };

// Deals the items of the tile's trees (tree l has s_items[l] of them, numbered 0..s_items[l]-1) evenly over the block's
// threads: used once per launch, for the nodes that exist when the kernel starts (the simulations keep a work list).  Every wave derives the same prefix sums from s_items in LDS and keeps them in its lanes 0..TM-1.
struct PoolScan { int mine, incl, total; };

template <int TM>
__device__ __forceinline__ PoolScan pool_scan(const int *s_items, int lane)
{
    PoolScan p;
    p.mine = lane < TM ? s_items[lane] : 0;
    p.incl = p.mine;
#pragma unroll
    for (int off = 1; off < TM; off <<= 1) {
        const int v = __shfl_up(p.incl, off, 64);
        p.incl += lane >= off ? v : 0;
    }
    p.total = __shfl(p.incl, TM - 1, 64);
    return p;
}

// The tree that owns item number `item` and the item's index inside that tree.  Call with every lane of the wave
// active (lane reads); lanes whose item is beyond the total get a harmless answer.
template <int TM>
__device__ __forceinline__ void pool_locate(const PoolScan &p, int item, int *tree, int *j)
{
    int lo = 0, hi = TM - 1;   // smallest tree whose inclusive prefix exceeds `item`
#pragma unroll
    for (int step = 0; step < Log2<TM>::value; step++) {
        const int mid = (lo + hi) >> 1;
        const int v = __shfl(p.incl, mid, 64);
        if (v > item) hi = mid; else lo = mid + 1;
    }
    const int tr = lo < TM ? lo : TM - 1;
    *tree = tr;
    *j = item - (__shfl(p.incl, tr, 64) - __shfl(p.mine, tr, 64));
}

// The records of a wave's 64 work-list nodes, fetched so that the memory pipeline sees whole cache lines: with one lane
// per node every 16-byte load of a wave touches 64 different lines (64 tag look-ups per instruction: the re-decision
// phase spent half its time there).  Here 8 lanes read 128 consecutive bytes of one record, 8 records per instruction,
// the wave parks them in LDS and every lane reads its own record's chunks back; the record's 128-byte pieces (one to
// four of them) take turns in the same staging area, the loads of the next piece in flight while this one is read back.
// `stage`: MZ_GC_STAGE bytes of LDS owned by the wave (the tile's activations, idle in this phase: the caller zeroes
// them again afterwards), `off`: byte offset of the lane's record from `base`, negative = no record.
#define MZ_GC_STAGE (64 * 144 + 256)   // 64 pieces of 128 bytes, 16 bytes apart (conflict-free 16-byte reads) + 64 offsets
template <int A, int IW>
__device__ __forceinline__ void rec_fetch_staged(GumbelRec<A, IW> &r, const char *base, int off, char *stage, int lane,
                                                 long long *stamps = nullptr)
{
#ifdef MZ_STAMPS
    const long long ft0 = clock64();
#endif
    constexpr int NCH = GumbelRec<A, IW>::NCH;
    constexpr int NP = (NCH + 7) / 8;                         // 128-byte pieces that hold fields
    constexpr int REC_CH = GumbelLayout<A, IW>::rec / 16;      // 16-byte chunks of a whole record
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // a native vector: arrays of it become registers
    int32_t *offs = reinterpret_cast<int32_t *>(stage + 64 * 144);
    u32x4 *st = reinterpret_cast<u32x4 *>(stage);
    offs[lane] = off;   // lanes without a node pass the offset of any record: loaded, never used
    lds_fence();
    const int sub = lane & 7, grp = lane >> 3;
    int so[8];
    u32x4 ln[8];
    static_for<8>([&](auto K) { constexpr int k = decltype(K)::value; so[k] = offs[8 * k + grp] + 16 * sub; });
    // chunk 8 p + sub of the record; the last piece of a record that ends inside it reads chunk 0 again instead of the
    // bytes behind the record (the last record of the last tree has nothing behind it)
    auto fetch = [&](auto P) {
        constexpr int p = decltype(P)::value;
        const int shift = (REC_CH >= 8 * p + 8 || 8 * p + sub < REC_CH) ? 128 * p : -16 * sub;
        static_for<8>([&](auto K) { constexpr int k = decltype(K)::value; ln[k] = *reinterpret_cast<const u32x4 *>(base + so[k] + shift); });
    };
    fetch(std::integral_constant<int, 0>());
    __builtin_amdgcn_sched_barrier(0);   // all eight loads in flight before the first store waits for its data
#ifdef MZ_STAMPS
    if (stamps) { stamps[0] += clock64() - ft0; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamps[1] += clock64() - ft0; }
#endif
    static_for<NP>([&](auto P) {
        constexpr int p = decltype(P)::value;
        static_for<8>([&](auto K) { constexpr int k = decltype(K)::value; st[(8 * k + grp) * 9 + sub] = ln[k]; });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (p + 1 < NP) {   // the next piece is on its way during the reads below
            fetch(std::integral_constant<int, p + 1>());
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_fence();
        constexpr int n = NCH - 8 * p < 8 ? NCH - 8 * p : 8;
        static_for<n>([&](auto K) { constexpr int k = decltype(K)::value; r.ch[8 * p + k] = __builtin_bit_cast(uint4, st[lane * 9 + k]); });
        lds_fence();
    });
#ifdef MZ_STAMPS
    if (stamps) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamps[2] += clock64() - ft0 + (long long)(r.ch[NCH - 1].x & 0); }
#endif
}

// An entry of the re-decision work list: tree of the tile (5 bits) | node (22 bits).
__device__ __forceinline__ int item_word(int tree, int node) { return (int)((unsigned)tree << 27 | (unsigned)node); }

// Dynamic LDS: activations [TM][stride] floats (at least one record staging area per wave) | decisions int32 [TM][N] |
// paths int32 [TM][S] | items int32 [TM][S + 1].
// Two waves per SIMD (256 registers): one 32-tree tile or two 16-tree tiles per compute unit.  (Three 16-tree tiles, 168
// registers each, only fit in LDS for the shortest searches and were measured slower: the decisions then spill.)
template <int TM, int A, int IW>
__global__ __launch_bounds__(TM * 16, 2) void k_search_gcache(Dev d, MfmaNet net, int first_leaf, int n_sims)
{
    extern __shared__ __align__(16) float act[];
    __shared__ __align__(16) double s_bk[TM * 16];
    __shared__ __align__(16) double s_gum[TM * A];
    __shared__ unsigned long long s_legal[TM];
    __shared__ int s_cvoff[TM], s_items[TM], s_total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = tid / 16, a = tid % 16, t = blockIdx.x * TM + i;
    const int H = net.io[6], N = d.L.N, S = d.L.S;
    const bool live = t < d.B, on = a < A;
    float *row = act + i * net.stride;
    // (a narrow network's activations are smaller than the record staging of the re-decision phase, which borrows them)
    constexpr int stage_floats = (TM / 4) * MZ_GC_STAGE / 4;
    const int act_floats = TM * net.stride > stage_floats ? TM * net.stride : stage_floats;
    int32_t *dec_all = reinterpret_cast<int32_t *>(act + act_floats);
    int32_t *path_all = dec_all + TM * N;
    int32_t *item_list = path_all + TM * S;   // the nodes to re-decide after a simulation (item_word below)
    double *bk = s_bk + i * 16;
    tile_zero<TM>(act, net.stride, tid);
    const int tt = live ? t : 0;  // groups past the end of the batch only keep the barriers company
    TreeMem m = tree_mem(d, tt);
    m.path = path_all + i * S;
    int32_t *dec = dec_all + i * N;
    float *hid = d.hidden + (long long)tt * N * H;
    if (on) s_gum[i * A + a] = d.gumbel[(long long)tt * A + a];
    if (a == 0) {
        s_legal[i] = d.legal[tt];
        s_cvoff[i] = d.m_considered[tt] * S;
        s_items[i] = live ? first_leaf : 0;
    }
    long long visited = 0;
    const NetBufs nb = net_bufs(net);
    TileSched ts;
    tile_sched_load<TM>(net, wave, lane, ts);
    __syncthreads();
    const char *tile_nodes = d.nodes + (long long)blockIdx.x * TM * d.L.tree_stride;
    // every node that already exists (a fresh search: the roots)
    {
        const PoolScan ps = pool_scan<TM>(s_items, lane);
        for (int base = 0; base < ps.total; base += TM * 16) {   // uniform trip count: pool_locate needs every lane
            int tree, node;
            pool_locate<TM>(ps, base + tid, &tree, &node);
            if (base + tid < ps.total) {
                GumbelRec<A, IW> r;
                rec_load<A, IW>(r, tile_nodes + (long long)tree * d.L.tree_stride + (long long)node * GumbelLayout<A, IW>::rec, ~0u);
                dec_all[tree * N + node] = gumbel_decide<A, IW>(d, r, node == 0, s_legal[tree], s_gum + tree * A,
                                                            d.cv_table + s_cvoff[tree]);
            }
        }
    }
    __syncthreads();
#ifdef MZ_STAMPS
    long long st_a = 0, st_b = 0, st_c = 0, st_d = 0, st_e = 0, st_t = clock64();
    long long sx[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sx_t = 0;   // finer: walk | epilogue | expansion+backup | fence | (barrier) | decision loads
#define MZ_XSTAMP(k) { long long n_ = clock64(); sx[k] += n_ - sx_t; sx_t = n_; }
#define MZ_XSTART() { sx_t = clock64(); }
#else
#define MZ_XSTAMP(k)
#define MZ_XSTART()
#endif
    for (int s = 0; s < n_sims; s++) {
        const int leaf = first_leaf + s;
        int parent = 0, action = 0, ntp = 0, depth = 0;
        BackupPre bp{0, 0, 0, 0, 0.0, 0.0};
        // (the first job's weights depend on nothing a simulation does: asked for before the walk, they have arrived
        // when the network begins)
        WReg wf;
        tile_wload(nb, ts, 0, lane, wf);
        MZ_XSTART()
        if (live) {
            // MCTS.simulate, mcts/mcts.py:98-114, over the decisions; all lanes of the group read the same words
            int node = 0;
            while (true) {
                const int w = dec[node];
                action = w & 0xff;
                const int child = (w >> 8) - 1;
                if (a == 0 && depth < S) m.path[depth] = (node << 6) | action;
                depth++;
                if (child < 0 || depth > S) break;  // depth bound: a malformed tree cannot spin forever
                node = child;
            }
            parent = node;
            ntp = d.players == 1 ? 0 : (depth & 1);  // starts at player 0, flips once per level (mcts/mcts.py:105,112)
            visited += depth + 1;
            if (a == 0) s_items[i] = depth < S ? depth : S;   // path nodes whose records this simulation changes
            MZ_XSTAMP(0)
            tile_stage_inputs(net, row, hid + (long long)parent * H, action, a);
            bp = backup_prefetch<16>(d, m, a, depth);   // the path's records travel while the network runs
        }
        MZ_MSTAMP(st_a)
        lds_barrier();
        MZ_MSTAMP(st_b)
        // The work list of this simulation's re-decision phase, made while nothing waits for it: every group puts its
        // path's nodes behind those of the groups before it (exclusive prefix of the path lengths by a 16-lane sum), so
        // the phase itself starts with one LDS read per thread instead of a search through prefix sums.
        // The roots come first (the live trees are the tile's first ones, and every path starts at its root): the root
        // rule and the interior rule are different code, and a wave whose lanes are all of one kind runs only that one.
        {
            const int n_live = d.B - (int)blockIdx.x * TM < TM ? d.B - (int)blockIdx.x * TM : TM;
            int before = 0;
            for (int k = a; k < TM; k += 16) before += (k < i && s_items[k] > 0) ? s_items[k] - 1 : 0;
            before = n_live + gsumi<16>(before);
            const int np = live ? (depth < S ? depth : S) : 0;
            if (a == 0 && np > 0) item_list[i] = item_word(i, 0);
            for (int j = 1 + a; j < np; j += 16) item_list[before + j - 1] = item_word(i, m.path[j] >> 6);
            if (tid == TM * 16 - 1) s_total = before + (np > 0 ? np - 1 : 0);
        }
        tile_run_levels<TM>(net, nb, act, ts, lane, wf);
        MZ_MSTAMP(st_c)
        MZ_XSTART()
        if (live) {
            // value and reward first: the backup needs them, and its record stores then travel while the new hidden
            // state is scaled and stored (below); the re-decision phase waits for the records, not for the state
            double value, reward;
            support_to_scalar2<16>(row + net.io[5], row + net.io[3], net.io[9], a, &value, &reward);
            const float lg = on ? row[net.io[4] + a] : 0.0f;
#ifdef MZ_STAMPS
            if (value == 12345.678 && lg == 7.0f) st_a++;
            MZ_XSTAMP(1)
#endif
            expand_backup<16, true, true>(d, m, a, on, parent, action, ntp, depth, leaf, lg, value, reward, bk, &bp);
            // The new leaf's decision needs no record: nothing below it is visited, every completed value is the
            // leaf's raw value, the value term of the interior rule is scale * ((raw - raw) / 1e-8) = 0 and the scores
            // are exp(logit) / sum - 0 (mcts/action_selection.py:99-118, mcts/qtransforms.py:25-48): the first largest
            // logit.  exp and the division by a common sum keep a lead of 1e-4 between logits strict (a factor
            // 1 + 1e-4 against errors of a few ulp), so with such a lead, a finite raw value and exp(best) neither 0
            // nor infinite the decision is written here; otherwise the leaf joins the work list.
            {
                const float ninf = -__builtin_huge_valf();
                const float l1 = on ? lg : ninf;
                const int nans = gsumi<16>((on && lg != lg) ? 1 : 0);
                const float m1 = gmaxf<16>(l1);
                const int best = gargmax<16>((double)l1, on ? a : 64);
                const float m2 = gmaxf<16>((on && a != best) ? lg : ninf);
                const bool sure = nans == 0 && (m1 - m2) > 1e-4f && m1 < 700.0f && m1 > -700.0f && (value - value) == 0.0;
                if (a == 0) {
                    if (sure) dec[leaf] = best;   // children_index is -1 on a fresh record: (-1 + 1) << 8 | action
                    else item_list[atomicAdd(&s_total, 1)] = item_word(i, leaf);
                }
            }
            tile_hidden_out<16>(net, row, a, hid + (long long)leaf * H);
#ifdef MZ_STAMPS
            MZ_XSTAMP(2)
#endif
        }
        __threadfence_block();   // the records the backup wrote are read by other lanes of the block below
#ifdef MZ_STAMPS
        MZ_XSTAMP(3)
#endif
        __syncthreads();
#ifdef MZ_STAMPS
        MZ_XSTAMP(4)
#endif
        MZ_MSTAMP(st_d)
        // the nodes whose records changed: the work list (path nodes, and leaves without a certified decision), one
        // lane per node
        {
#ifdef MZ_STAMPS
            const long long ph_t0 = clock64();
#endif
            const int total = s_total;
            char *stage = reinterpret_cast<char *>(act) + wave * MZ_GC_STAGE;
            for (int base = 0; base < total; base += TM * 16) {
                const bool has = base + tid < total;
                const int w = has ? item_list[base + tid] : 0;
                const int tree = (int)((unsigned)w >> 27), node = w & 0x3fffff;
                GumbelRec<A, IW> grec;
                if (base + (wave << 6) >= total) continue;   // wave-uniform: none of the wave's lanes has a node
                #ifdef MZ_STAMPS
                rec_fetch_staged<A, IW>(grec, tile_nodes, has ? tree * (int)d.L.tree_stride + node * GumbelLayout<A, IW>::rec : 0, stage, lane, &sx[12]);
#else
                rec_fetch_staged<A, IW>(grec, tile_nodes, has ? tree * (int)d.L.tree_stride + node * GumbelLayout<A, IW>::rec : 0, stage, lane);
#endif
                if (has) {
#ifdef MZ_STAMPS
                    sx[11] += clock64() - ph_t0 + node * 0;
                    dec_all[tree * N + node] = gumbel_decide<A, IW>(d, grec, node == 0, s_legal[tree], s_gum + tree * A,
                                                                d.cv_table + s_cvoff[tree], &sx[5]);
#else
                    dec_all[tree * N + node] = gumbel_decide<A, IW>(d, grec, node == 0, s_legal[tree], s_gum + tree * A,
                                                                d.cv_table + s_cvoff[tree]);
#endif
                }
            }
            if ((wave << 6) < total) {   // the staging area goes back to the network as zeros
                uint4 *st = reinterpret_cast<uint4 *>(stage);
                for (int k = lane; k < MZ_GC_STAGE / 16; k += 64) st[k] = uint4{0, 0, 0, 0};
            }
        }
        __syncthreads();
        MZ_MSTAMP(st_e)
    }
#ifdef MZ_STAMPS
    if (live && a == 0) {  // diagnostic build only: [walk+stage, barrier wait, network, epilogue+backup, decisions] cycles
        long long *o = reinterpret_cast<long long *>(d.path + (long long)t * d.L.S);
        o[0] = st_a; o[1] = st_b; o[2] = st_c; o[3] = st_d; o[4] = st_e;
        for (int k = 0; k < 16; k++) o[5 + k] = sx[k];
    }
#endif
    if (live && a == 0) d.visit_total[t] += visited;
}

}  // namespace mz
