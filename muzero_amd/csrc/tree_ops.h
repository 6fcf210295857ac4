// Per-tree search steps executed by one lane group: descent (selection), expansion, backup,
// and the final action rule.  Arithmetic, operand order and dtypes follow the reference's
// NumPy expressions (cited per function) so every tree field comes out bit-identical.
#pragma once
#include "device_common.h"

namespace mz {

// A row of small integers (visit counts, node indices) stored as int32 or int16.
struct IntRow {
    char *p;
    bool i16;
    __device__ __forceinline__ int get(int a) const
    {
        return i16 ? (int)reinterpret_cast<const int16_t *>(p)[a] : reinterpret_cast<const int32_t *>(p)[a];
    }
    __device__ __forceinline__ void set(int a, int v) const
    {
        if (i16) reinterpret_cast<int16_t *>(p)[a] = (int16_t)v;
        else reinterpret_cast<int32_t *>(p)[a] = v;
    }
};

struct RowView {
    NodeHdr *hdr;
    double *cval, *crew;
    void *prior;
    IntRow cvis, cidx;
};

__device__ __forceinline__ RowView view(const Dev &d, const TreeMem &m, int n)
{
    char *r = m.nodes + n * d.L.rec;
    RowView v;
    v.hdr = reinterpret_cast<NodeHdr *>(r);
    v.cval = reinterpret_cast<double *>(r + d.L.off_cval);
    v.crew = reinterpret_cast<double *>(r + d.L.off_crew);
    v.prior = r + d.L.off_prior;
    v.cvis = IntRow{r + d.L.off_cvis, d.L.idx16 != 0};
    v.cidx = IntRow{r + d.L.off_cidx, d.L.idx16 != 0};
    return v;
}

struct NodeStats {  // what lane `a` holds of the node being scored
    int nv;           // node_visits[node]
    double nval, raw; // node_values[node], node_raw_values[node]
    int cv, ci;       // children_visits[node, a], children_index[node, a]
    double prior;     // children_prior_logits[node, a] (exact: f32 priors widen losslessly)
    double r, v;      // children_rewards[node, a], children_values[node, a]
};

template <bool GUMBEL>
__device__ __forceinline__ NodeStats load_node(const Dev &d, const TreeMem &m, int node, int a, bool on)
{
    RowView w = view(d, m, node);
    NodeStats s;
    s.nv = w.hdr->visits;
    s.nval = w.hdr->value;
    s.raw = w.hdr->raw;
    s.cv = 0; s.ci = -1; s.prior = 0; s.r = 0; s.v = 0;
    if (on) {
        s.cv = w.cvis.get(a);
        s.ci = w.cidx.get(a);
        s.prior = GUMBEL ? (double)reinterpret_cast<float *>(w.prior)[a]
                         : reinterpret_cast<double *>(w.prior)[a];
        s.r = w.crew[a];
        s.v = w.cval[a];
    }
    return s;
}

// mcts/qtransforms.py:7-22 (with Tree.qvalues, mcts/tree.py:55-58)
template <int LANES>
__device__ __forceinline__ double qt_parent_siblings(const Dev &d, const NodeStats &s, bool on)
{
    const double inf = __builtin_huge_val();
    double q = s.r + d.discount * s.v;
    double safe = s.cv > 0 ? q : s.nval;
    double lo = gmin<LANES>(on ? safe : inf), hi = gmax<LANES>(on ? safe : -inf);
    lo = s.nval < lo ? s.nval : lo;
    hi = s.nval > hi ? s.nval : hi;
    double den = hi - lo;
    den = den < 1e-8 ? 1e-8 : den;  // np.maximum: a NaN stays a NaN
    double cq = s.cv > 0 ? q : lo;
    return (cq - lo) / den;
}

// mcts/qtransforms.py:51-77: the per-action vector the reference's compute_mixed_value yields.
template <int LANES>
__device__ __forceinline__ double mixed_value(const Dev &d, const NodeStats &s, int a, bool on,
                                              double q, int vsum)
{
    const int A = d.L.A;
    float e = on ? np_expf((float)s.prior) : 0.0f;
    float es = gsum_np<LANES, float>(e, a, A);
    float p = __fdiv_rn(e, es);
    const float tiny = 1.17549435e-38f;
    p = p < tiny ? tiny : p;  // np.maximum(tiny, p): a NaN (0 / 0 when every exp underflows) stays a NaN
    float pm = (on && s.cv > 0) ? p : 0.0f;
    float prob_sum = gsum_np<LANES, float>(pm, a, A);
    double pq = (on && s.cv > 0) ? (double)p * q : 0.0;
    double sum_pq = gsum_np<LANES, double>(pq, a, A);
    float denf = s.cv > 0 ? prob_sum : 1.0f;
    double wq = sum_pq / (double)denf;
    return (s.raw + (double)vsum * wq) / (double)(vsum + 1);
}

// mcts/qtransforms.py:25-48
template <int LANES>
__device__ __forceinline__ double qt_completed_mix(const Dev &d, const NodeStats &s, int a, bool on,
                                                   int *vsum_out)
{
    const double inf = __builtin_huge_val();
    double q = s.r + d.discount * s.v;
    int vsum = gsumi<LANES>(on ? s.cv : 0);
    int vmax = gmaxi<LANES>(on ? s.cv : -2147483647);
    double mixed = mixed_value<LANES>(d, s, a, on, q, vsum);
    double comp = s.cv > 0 ? q : mixed;
    double lo = gmin<LANES>(on ? comp : inf), hi = gmax<LANES>(on ? comp : -inf);
    double den = hi - lo;
    den = den < 1e-8 ? 1e-8 : den;  // np.maximum: a NaN stays a NaN
    double scale = (d.c_visit + (double)vmax) * d.c_scale;
    *vsum_out = vsum;
    return scale * ((comp - lo) / den);
}

// mcts/action_selection.py:147-157
template <int LANES>
__device__ __forceinline__ double gumbel_score(const NodeStats &s, bool on, double g, double cq, int considered)
{
    const float ninf = -__builtin_huge_valf();
    float lmax = gmaxf<LANES>(on ? (float)s.prior : ninf);
    float l = (float)s.prior - lmax;
    double sc = (g + (double)l) + cq;
    sc = sc < -1e9 ? -1e9 : sc;  // np.maximum(-1e9, .) keeps a NaN
    return sc + (s.cv == considered ? 0.0 : -__builtin_huge_val());
}

// End of a kernel's descents: the tree's running count of PUCT selection calls.  A caller-supplied noise buffer
// (MZ_NOISE_BUFFER) shorter than the calls made is reported through the device error flag - the calls past its
// end were scored with u = 0, which is not what the caller asked for.
__device__ __forceinline__ void store_noise_calls(const Dev &d, int t, uint32_t calls, int noise_max_calls)
{
    d.noise_calls[t] = calls;
    if (d.noise_mode == 1 && (long long)calls > (long long)noise_max_calls) atomicExch(d.err, 2);
}

// Per-descent constants hoisted out of the level loop.
struct DescendCtx {
    unsigned long long legal;     // root legality mask of the tree
    unsigned long long key;       // counter-noise key of the tree
    const double *c_table;        // exploration constants in global memory
    const double *c_table_lds;    // LDS copy of the whole table when the kernel staged one, else NULL
    const double *sq_table;       // sqrt(n / (1 + c)) table in LDS, or NULL
    const double *sqrt_n;         // [S + 2] sqrt(n) and
    const double *rsqrt_1c;       // [S + 2] 1 / sqrt(1 + c) in LDS for the cached PUCT decisions, or NULL
    const int32_t *cv_row;        // sequential-halving schedule of this tree's considered count
    double gumbel_a;              // this lane's Gumbel noise
};

template <bool GUMBEL>
__device__ __forceinline__ DescendCtx descend_ctx(const Dev &d, int t, int a, bool on, const double *c_table_lds,
                                                  const double *sq_table_lds = nullptr)
{
    DescendCtx c;
    c.sq_table = sq_table_lds;
    c.sqrt_n = c.rsqrt_1c = nullptr;
    c.legal = d.legal[t];
    c.key = GUMBEL ? 0ull : noise_key(d.seed, d.tree_id0 + (unsigned long long)t);
    c.c_table = d.c_table;
    c.c_table_lds = c_table_lds;
    c.cv_row = GUMBEL ? d.cv_table + (long long)d.m_considered[t] * d.L.S : nullptr;
    c.gumbel_a = (GUMBEL && on) ? d.gumbel[(long long)t * d.L.A + a] : 0.0;
    return c;
}

// One selection call at `node` (mcts/action_selection.py:11-22 switch; :25-49 PUCT;
// :73-96 Gumbel root; :99-118 Gumbel interior).  `u` is this lane's tie-break uniform (PUCT).
// LANES here is the number of action lanes (RL) of the group; only those lanes call it.
template <int LANES, bool GUMBEL>
__device__ __forceinline__ int select_at(const Dev &d, const DescendCtx &c, const TreeMem &m, int node, int depth, int a,
                                         bool on, double u, int *child_out)
{
    const double inf = __builtin_huge_val();
    NodeStats s = load_node<GUMBEL>(d, m, node, a, on);
    bool legal = depth != 0 || ((c.legal >> a) & 1ull);
    double score;
    if constexpr (!GUMBEL) {
        double nq = qt_parent_siblings<LANES>(d, s, on);
        // two typed loads (LDS or global) instead of one through a generic pointer
        double cc = c.c_table_lds ? c.c_table_lds[s.nv] : c.c_table[s.nv];
        // sqrt(N / (1 + n_a)): integer arguments, so a table of the correctly rounded values is bit-identical
        double sq = c.sq_table ? c.sq_table[s.nv * (d.L.S + 1) + s.cv] : sqrt((double)s.nv / (double)(1 + s.cv));
        double uu = cc * s.prior + sq;
        score = (uu + nq) + 1e-7 * u;
    } else {
        int vsum;
        double cq = qt_completed_mix<LANES>(d, s, a, on, &vsum);
        if (depth == 0) {
            int considered = c.cv_row[vsum < d.L.S ? vsum : d.L.S - 1];
            score = gumbel_score<LANES>(s, on, c.gumbel_a, cq, considered);
        } else {
            double e = on ? exp(s.prior + cq) : 0.0;
            double es = gsum_np<LANES, double>(e, a, d.L.A);
            score = e / es - (double)s.cv / (double)(1 + vsum);
        }
    }
    // np.argmax treats a NaN as the maximum and returns the first one (a node whose float32 prior softmax
    // underflows entirely scores NaN everywhere in the reference: it then takes the first legal action)
    score = score != score ? inf : score;
    if (!on || !legal) score = -inf;
    int best = gargmax<LANES>(score, on ? a : 0x7fffffff);
    // the chosen action's child index sits in lane `best`; min over (lane == best ? ci : big) fetches it
    *child_out = -gmaxi<LANES>(a == best ? -s.ci : -0x7fffffff);
    return best;
}

// MCTS.simulate, mcts/mcts.py:98-114.  Returns the parent node; action/next_to_play/depth by reference.
// Records the path for the backup.  noise: MZ_NOISE_BUFFER rows for this tree, or NULL.
// The RL action lanes walk the tree; the result is then shared with the rest of the group.
// `calls`: PUCT selection calls made so far in this search (indexes the tie-break noise); advanced here
// on the action lanes.
template <int LANES, int RL, bool GUMBEL>
__device__ __forceinline__ int descend(const Dev &d, const DescendCtx &c, const TreeMem &m, int a, bool on,
                                       const double *noise, int noise_max_calls, uint32_t &calls, int *action_out,
                                       int *ntp_out, int *depth_out)
{
    int node = 0, depth = 0, ntp = 0, action = 0;
    if (LANES == RL || a < RL) {
        int32_t *path = m.path;
        while (true) {
            double u = 0.0;
            if constexpr (!GUMBEL) {
                if (on) {
                    if (d.noise_mode == 0) u = noise_from_key(c.key, calls, (unsigned)a);
                    else if (d.noise_mode == 1 && (int)calls < noise_max_calls)
                        u = noise[(long long)calls * d.L.A + a];
                }
                calls++;
            }
            int child;
            action = select_at<RL, GUMBEL>(d, c, m, node, depth, a, on, u, &child);
            if (a == 0 && depth < d.L.S) path[depth] = (node << 6) | action;
            ntp = d.players == 1 ? ntp : 1 - ntp;
            depth++;
            if (child < 0 || depth > d.L.S) break;  // depth bound: a malformed tree cannot spin forever
            node = child;
        }
    }
    if constexpr (RL < LANES) {  // lanes beyond the action lanes take lane 0's result
        int packed = (node << 6) | action;
        packed = gshfl<LANES>(packed, 0);
        int dn = gshfl<LANES>((depth << 1) | ntp, 0);
        node = packed >> 6; action = packed & 63; depth = dn >> 1; ntp = dn & 1;
    }
    *action_out = action; *ntp_out = ntp; *depth_out = depth;
    return node;
}

// ---------------------------------------------------------------------------------------------------
// Cached PUCT decisions.  The action muzero_action_selection picks at a node (mcts/action_selection.py:25-49 with
// mcts/qtransforms.py:7-22) is the argmax of u + q + 1e-7 * uniform[0, 1): u and q are functions of the node's own
// record, which only changes when the node lies on a simulation's path, and the tie-break term is below 1e-7.  Where
// the best u + q leads the runner-up by more than that, no draw can change the choice: the decision is kept in a table
// (word = (children_index[node, action] + 1) << 8 | action) and the next descents read it instead of scoring the node.
// Nodes without such a lead keep the word 0 and are scored in full, with the call's draws, whenever a descent meets
// them.  A descent still counts one selection call per level, so the noise stream stays where the reference's is.
//
// One LANE evaluates one node, looping over the actions.  The scores are evaluated cheaply (reciprocals refined by
// Newton steps instead of divisions and a square root): they differ from the reference's by a few units in the last
// place of the score (1e-15 relative), which the lead demanded, 1e-7 plus a millionth of that per unit of score, covers.
// NaN or infinite scores are never certified.
__device__ __forceinline__ double rcp_refined(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}

// G lanes share a node (lane g of the group takes actions g, g + G, ...): short paths are decided by wide groups, long
// ones by many narrow groups at once.  Every lane of the wave must call it (the group reductions move data across
// lanes); all lanes of a group return the node's word.
#define MZ_PUCT_NA 8   // actions per lane that puct_decide_regs keeps in registers
template <int G>
__device__ __forceinline__ int puct_decide(const Dev &d, const DescendCtx &c, const TreeMem &m, int node, int g)
{
    constexpr int UB = 4;   // actions per lane whose loads are issued together (a loop over single actions pays the
                            // memory latency once per action)
    const RowView w = view(d, m, node);
    const int A = d.L.A, nv = w.hdr->visits;
    const double nval = w.hdr->value, inf = __builtin_huge_val();
    const double *prior = reinterpret_cast<const double *>(w.prior);
    // an entry of the visit / index rows without a branch on their width: the aligned 32-bit word that holds it
    const bool i16 = w.cvis.i16;
    auto row_get = [&](const char *p, int a) {
        const int word = *reinterpret_cast<const int32_t *>(p + (i16 ? (2 * a) & ~3 : 4 * a));
        return i16 ? (int)(int16_t)(word >> ((a & 1) * 16)) : word;
    };
    double lo = nval, hi = nval;
    int bad = nval != nval;
    for (int a0 = g; a0 < A; a0 += UB * G) {
        double q[UB];
        int cv[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) {
            const int a = a0 + k * G < A ? a0 + k * G : g;   // past the end: the lane's first action again (harmless here)
            cv[k] = row_get(w.cvis.p, a);
            q[k] = w.crew[a] + d.discount * w.cval[a];
        }
#pragma unroll
        for (int k = 0; k < UB; k++) {
            const double safe = cv[k] > 0 ? q[k] : nval;
            bad |= safe != safe;
            lo = safe < lo ? safe : lo;
            hi = safe > hi ? safe : hi;
        }
    }
    lo = gmin<G>(lo); hi = gmax<G>(hi);
    double den = hi - lo;
    den = den < 1e-8 ? 1e-8 : den;
    const double rden = rcp_refined(den);
    const double cc = c.c_table_lds ? c.c_table_lds[nv] : c.c_table[nv];
    const double nvd = (double)nv;
    double b1 = -inf, b2 = -inf;
    int best = 0x7fffffff;
    for (int a0 = g; a0 < A; a0 += UB * G) {
        double q[UB], pr[UB];
        int cv[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) {
            const int a = a0 + k * G < A ? a0 + k * G : g;
            cv[k] = row_get(w.cvis.p, a);
            q[k] = w.crew[a] + d.discount * w.cval[a];
            pr[k] = prior[a];
        }
#pragma unroll
        for (int k = 0; k < UB; k++) {
            const int a = a0 + k * G;
            // masked_argmax at the root: illegal actions never win
            if (a >= A || (node == 0 && !((c.legal >> a) & 1ull))) continue;
            const double nq = ((cv[k] > 0 ? q[k] : lo) - lo) * rden;
            // sqrt(N / (1 + n_a)) = x * rsqrt(x), the reciprocal square root refined twice
            const double x = nvd * rcp_refined((double)(1 + cv[k]));
            double r = __builtin_amdgcn_rsq(x);
            r = r * __builtin_fma(-0.5 * x * r, r, 1.5);
            r = r * __builtin_fma(-0.5 * x * r, r, 1.5);
            const double sq = nv > 0 ? x * r : 0.0;
            const double sc = (cc * pr[k] + sq) + nq;
            bad |= sc != sc;
            if (sc > b1) { b2 = b1; b1 = sc; best = a; }
            else if (sc > b2) b2 = sc;
        }
    }
    // the group's best and runner-up: the runner-up is the largest of the other lanes' bests and the winner's own second;
    // two lanes with the same best are a tie (no lead)
    const double gb1 = gmax<G>(b1);
    const bool mine = b1 == gb1;
    const int winners = gsumi<G>(mine ? 1 : 0);
    double gb2 = gmax<G>(mine ? b2 : b1);
    gb2 = winners > 1 ? gb1 : gb2;
    const int gbest = -gmaxi<G>(mine ? -best : -0x7fffffff);
    bad = gsumi<G>(bad);
    const bool sure = !bad && gb1 > -1e300 && gb1 < 1e300 && (gb1 - gb2) > 1.000001e-7 + 1e-13 * __builtin_fabs(gb1);
    const int gsafe = gbest < A ? gbest : 0;
    return sure ? ((row_get(w.cidx.p, gsafe) + 1) << 8) | gbest : 0;
}

// The same decision when a lane's share of the actions fits in registers (at most NA per lane): every value the node
// needs - the child indices included - is asked for at once (puct_load), so a lane waits for memory once per node
// instead of once per batch and pass and once more for the winner's child; the caller asks for the next node's values
// before it evaluates this one's (puct_decide_regs).
template <int NA>
struct PuctRegs {
    double q[NA], pr[NA], nval;
    int cv[NA], ci[NA], nv;
};

template <int G, int NA>
__device__ __forceinline__ void puct_load(const Dev &d, const TreeMem &m, int node, int g, PuctRegs<NA> &r)
{
    const RowView w = view(d, m, node);
    const int A = d.L.A;
    const double *prior = reinterpret_cast<const double *>(w.prior);
    const bool i16 = w.cvis.i16;
    auto row_get = [&](const char *p, int a) {   // an entry of the visit / index rows without a branch on their width
        const int word = *reinterpret_cast<const int32_t *>(p + (i16 ? (2 * a) & ~3 : 4 * a));
        return i16 ? (int)(int16_t)(word >> ((a & 1) * 16)) : word;
    };
    r.nv = w.hdr->visits;
    r.nval = w.hdr->value;
#pragma unroll
    for (int k = 0; k < NA; k++) {
        // past the end: an action that exists, again (it enters the range twice, which changes nothing, and is never
        // scored; a lane whose first action is already past the end - groups wider than the action count - takes action 0)
        const int a = g + k * G < A ? g + k * G : (g < A ? g : 0);
        r.cv[k] = row_get(w.cvis.p, a);
        r.ci[k] = row_get(w.cidx.p, a);
        r.q[k] = w.crew[a] + d.discount * w.cval[a];
        r.pr[k] = prior[a];
    }
}

template <int G, int NA>
__device__ __forceinline__ int puct_decide_regs(const Dev &d, const DescendCtx &c, int node, int g, const PuctRegs<NA> &r)
{
    const int A = d.L.A, nv = r.nv;
    const double nval = r.nval, inf = __builtin_huge_val();
    const double cc = c.c_table_lds ? c.c_table_lds[nv] : c.c_table[nv];
    double lo = nval, hi = nval;
    int bad = nval != nval;
#pragma unroll
    for (int k = 0; k < NA; k++) {
        const double safe = r.cv[k] > 0 ? r.q[k] : nval;
        bad |= safe != safe;
        lo = safe < lo ? safe : lo;
        hi = safe > hi ? safe : hi;
    }
    lo = gmin<G>(lo); hi = gmax<G>(hi);
    double den = hi - lo;
    den = den < 1e-8 ? 1e-8 : den;
    const double rden = rcp_refined(den);
    const double nvd = (double)nv, sqn = c.sqrt_n ? c.sqrt_n[nv] : 0.0;
    double b1 = -inf, b2 = -inf;
    int best = 0x7fffffff, best_ci = -1;
#pragma unroll
    for (int k = 0; k < NA; k++) {
        const int a = g + k * G;
        if (a >= A || (node == 0 && !((c.legal >> a) & 1ull))) continue;   // masked_argmax at the root
        const double nq = ((r.cv[k] > 0 ? r.q[k] : lo) - lo) * rden;
        double sq;   // sqrt(N / (1 + n_a)) to a few units in the last place
        if (c.sqrt_n) {
            sq = sqn * c.rsqrt_1c[r.cv[k]];
        } else {
            const double x = nvd * rcp_refined((double)(1 + r.cv[k]));
            double rs = __builtin_amdgcn_rsq(x);
            rs = rs * __builtin_fma(-0.5 * x * rs, rs, 1.5);
            rs = rs * __builtin_fma(-0.5 * x * rs, rs, 1.5);
            sq = nv > 0 ? x * rs : 0.0;
        }
        const double sc = (cc * r.pr[k] + sq) + nq;
        bad |= sc != sc;
        if (sc > b1) { b2 = b1; b1 = sc; best = a; best_ci = r.ci[k]; }
        else if (sc > b2) b2 = sc;
    }
    const double gb1 = gmax<G>(b1);
    const bool mine = b1 == gb1;
    const int winners = gsumi<G>(mine ? 1 : 0);
    const double gb2 = gmax<G>(mine ? b2 : b1);
    const int gbest = -gmaxi<G>(mine ? -best : -0x7fffffff);
    const int gchild1 = gmaxi<G>(mine ? best_ci + 1 : 0);   // children_index + 1 of the winner (one winner: see `sure`)
    bad = gsumi<G>(bad);
    const bool sure = !bad && winners == 1 && gb1 > -1e300 && gb1 < 1e300 &&
                      (gb1 - gb2) > 1.000001e-7 + 1e-13 * __builtin_fabs(gb1);
    return sure ? (gchild1 << 8) | gbest : 0;
}

// The decisions of nodes [0, n) of a list (`node_of(j)` names the j-th) into `dec` and, when the wave keeps a copy of
// its tree's table in LDS, into that copy (`dec_lds`, else NULL), by the whole wave.
template <int G, typename F>
__device__ __forceinline__ void puct_decide_list(const Dev &d, const DescendCtx &c, const TreeMem &m, int32_t *dec,
                                                 int32_t *dec_lds, int n, int lane, F node_of)
{
    constexpr int PER = 64 / G;   // nodes per pass
    const int g = lane % G;
    auto put = [&](bool has, int node, int word) {
        if (has && g == 0) {
            dec[node] = word;
            if (dec_lds) dec_lds[node] = word;
        }
    };
    const int per_lane = (d.L.A + G - 1) / G;
    auto pipelined = [&](auto na) {
        constexpr int NA = decltype(na)::value;
        PuctRegs<NA> cur, nxt;
        int j = lane / G;
        bool has = j < n;
        int node = has ? node_of(j) : 0;
        puct_load<G>(d, m, node, g, cur);
        for (int j0 = 0; j0 < n; j0 += PER) {   // wave-uniform trip count
            const int jn = j + PER;
            const bool has_n = jn < n;
            const int node_n = has_n ? node_of(jn) : 0;
            if (j0 + PER < n) puct_load<G>(d, m, node_n, g, nxt);   // on its way while this pass is evaluated
            put(has, node, puct_decide_regs<G>(d, c, node, g, cur));
            cur = nxt; j = jn; has = has_n; node = node_n;
        }
    };
    if (per_lane <= 4) {
        pipelined(std::integral_constant<int, 4>());
    } else if (per_lane <= 6) {
        pipelined(std::integral_constant<int, 6>());
    } else if (per_lane <= MZ_PUCT_NA) {
        pipelined(std::integral_constant<int, MZ_PUCT_NA>());
    } else {
        for (int j0 = 0; j0 < n; j0 += PER) {
            const int j = j0 + lane / G;
            const bool has = j < n;
            const int node = has ? node_of(j) : 0;
            put(has, node, puct_decide<G>(d, c, m, node, g));
        }
    }
}

// descend() over the decision table of a whole-wave group: a table hit costs one 4-byte read per level - from the wave's
// LDS copy of the table (`dec_lds`) when it has one, else from memory, where every level of the walk waits for a trip
// to HBM; a miss is scored by the RL action lanes exactly as descend() does.  Every lane takes part; results are
// wave-uniform.
// The path goes to memory 64 levels at a time (lane k keeps level k of the current stretch): a store per level would make
// every level wait for the one before it, since the wait for a table word from memory also covers the stores in flight.
template <int RL, bool IN_LDS>
__device__ __forceinline__ int descend_cached_puct(const Dev &d, const DescendCtx &c, const TreeMem &m, const int32_t *dec,
                                                   const int32_t *dec_lds, int a, bool on, const double *noise,
                                                   int noise_max_calls, uint32_t &calls, int *action_out, int *ntp_out,
                                                   int *depth_out, int *misses = nullptr)
{
    int node = 0, depth = 0, ntp = 0, action = 0, path_reg = 0;
    while (true) {
        // (LDS reads and atomic loads of workgroup scope: the compiler neither moves them into the scalar cache, which
        // does not see this kernel's stores, nor sends them past the L2 as it does volatile ones)
        int word;
        if constexpr (IN_LDS) word = __builtin_amdgcn_readfirstlane(dec_lds[node]);
        else word = __builtin_amdgcn_readfirstlane(__hip_atomic_load(dec + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        int child;
        if (word != 0) {
            action = word & 0xff;
            child = (word >> 8) - 1;
        } else {
            if (misses) *misses += 1;
            int packed = 0;
            if (a < RL) {
                double u = 0.0;
                if (on) {
                    if (d.noise_mode == 0) u = noise_from_key(c.key, calls, (unsigned)a);
                    else if (d.noise_mode == 1 && (int)calls < noise_max_calls) u = noise[(long long)calls * d.L.A + a];
                }
                int ch;
                const int act = select_at<RL, false>(d, c, m, node, depth, a, on, u, &ch);
                packed = ((ch + 1) << 8) | act;
            }
            packed = __builtin_amdgcn_readfirstlane(packed);
            action = packed & 0xff;
            child = (packed >> 8) - 1;
        }
        calls++;
        path_reg = a == (depth & 63) ? (node << 6) | action : path_reg;
        ntp = d.players == 1 ? ntp : 1 - ntp;
        depth++;
        if ((depth & 63) == 0 && depth - 64 + a < d.L.S) m.path[depth - 64 + a] = path_reg;   // a full stretch
        if (child < 0 || depth > d.L.S) break;
        node = child;
    }
    if ((depth & 63) != 0 && a < (depth & 63) && (depth & ~63) + a < d.L.S) m.path[(depth & ~63) + a] = path_reg;
    *action_out = action; *ntp_out = ntp; *depth_out = depth;
    return node;
}

// The same descent for a whole-wave group over a tree of at most 2 * (64 / RL) nodes whose records sit
// in LDS: every live node's selection is evaluated at once (RL lanes per node; the decision at a node is
// a pure function of its statistics and of the tie-break noise of call `calls + depth(node)`), then the
// path is read off the per-node results with lane reads.  Same results as descend().
template <int RL, bool GUMBEL>
__device__ __forceinline__ int descend_wide(const Dev &d, const DescendCtx &c, const TreeMem &m, int lane, int n_live,
                                            const double *noise, int noise_max_calls, uint32_t &calls,
                                            int *action_out, int *ntp_out, int *depth_out)
{
    constexpr int NP = 64 / RL;  // nodes per pass
    const int a = lane % RL, slot = lane / RL;
    const bool on = a < d.L.A;
    int pack[2] = {0, 0};        // per node: (child + 1) << 8 | action, held by the node's lanes
#pragma unroll
    for (int p = 0; p < 2; p++) {
        if (p * NP >= n_live) break;
        int node = p * NP + slot;
        const bool live = node < n_live;
        node = live ? node : 0;  // spare lanes redo the root so that every load stays on initialised records
        const int dn = m.depth[node];
        double u = 0.0;
        if constexpr (!GUMBEL) {
            const uint32_t call = calls + (uint32_t)dn;
            if (on) {
                if (d.noise_mode == 0) u = noise_from_key(c.key, call, (unsigned)a);
                else if (d.noise_mode == 1 && (int)call < noise_max_calls) u = noise[(long long)call * d.L.A + a];
            }
        }
        int child;
        int act = select_at<RL, GUMBEL>(d, c, m, node, dn, a, on, u, &child);
        pack[p] = ((child + 1) << 8) | act;
    }
    int node = 0, depth = 0, ntp = 0, action = 0;
    while (true) {
        int pk = node < NP ? __builtin_amdgcn_readlane(pack[0], node * RL)
                           : __builtin_amdgcn_readlane(pack[1], (node - NP) * RL);
        action = pk & 0xff;
        int child = (pk >> 8) - 1;
        if (lane == 0 && depth < d.L.S) m.path[depth] = (node << 6) | action;
        ntp = d.players == 1 ? ntp : 1 - ntp;
        depth++;
        if (child < 0 || child >= n_live || depth > d.L.S) break;
        node = child;
    }
    calls += (uint32_t)depth;
    *action_out = action; *ntp_out = ntp; *depth_out = depth;
    return node;
}

// MCTS.expand (mcts/mcts.py:116-141) for leaf index `leaf`, then MCTS.backward (:143-164) along the
// recorded path.  `logit` is this lane's prior logit; value/reward are group-uniform.
// `lds_r`: LANES doubles of LDS owned by this group (staging of the per-level rewards).
// What the backup reads of the path's first LANES levels (lane i: level i above the leaf).  A kernel that knows the
// path long before it knows the leaf's value (the leaf is evaluated in between) asks for these early
// (backup_prefetch) and hands them to expand_backup: the records then travel from HBM while the network runs.
struct BackupPre {
    int pk, nv, tp, cvs;
    double V, r;
};

template <int LANES>
__device__ __forceinline__ BackupPre backup_prefetch(const Dev &d, const TreeMem &m, int a, int depth)
{
    BackupPre p{0, 0, 0, 0, 0.0, 0.0};
    if (a < depth) {
        p.pk = m.path[depth - 1 - a];
        const RowView pw = view(d, m, p.pk >> 6);
        p.nv = pw.hdr->visits; p.V = pw.hdr->value; p.tp = pw.hdr->to_play;
        p.r = pw.crew[p.pk & 63];
        p.cvs = pw.cvis.get(p.pk & 63);
    }
    return p;
}

template <int LANES, bool GUMBEL, bool PRE = false>
__device__ __forceinline__ void expand_backup(const Dev &d, const TreeMem &m, int a, bool on, int parent, int action,
                                              int ntp, int depth, int leaf, float logit, double value,
                                              double reward, double *lds_r, const BackupPre *pre = nullptr)
{
    // --- expansion: a fresh record for the leaf, links in the parent
    RowView lf = view(d, m, leaf);
    if (a == 0) {
        NodeHdr h;
        h.visits = 1; h.parent = parent; h.afp = action; h.to_play = ntp; h.value = value; h.raw = value;
        *lf.hdr = h;
        if (m.depth) m.depth[leaf] = depth;
        RowView pr = view(d, m, parent);
        pr.cidx.set(action, leaf);
        pr.crew[action] = reward;
    }
    if (on) {
        lf.cval[a] = 0.0; lf.crew[a] = 0.0; lf.cvis.set(a, 0); lf.cidx.set(a, -1);
        if (GUMBEL) reinterpret_cast<float *>(lf.prior)[a] = logit;
        else reinterpret_cast<double *>(lf.prior)[a] = (double)logit;
    }
    // --- backup: lane i of a chunk owns level i above the leaf; loads are independent, only the
    // discounted return L is a serial recurrence (reward + gamma * L, mcts/mcts.py:162), which every
    // lane replays from the rewards staged in LDS (reads run ahead of the dependent multiply-add chain).
    const int32_t *path = m.path;
    double L = value, carry = value;
    for (int base = 0; base < depth; base += LANES) {
        int i = base + a;
        bool valid = i < depth;
        int p = 0, pa = 0, nv = 0, tp = 0, cvs = 0;
        double V = 0, r = 0;
        RowView pw;
        if (valid) {
            if (PRE && base == 0) {   // read ahead by the caller, before the expansion wrote the parent's link and
                p = pre->pk >> 6; pa = pre->pk & 63;   // reward: level 0's reward is the leaf's, not the stored one
                pw = view(d, m, p);
                nv = pre->nv; V = pre->V; tp = pre->tp; cvs = pre->cvs;
                r = i == 0 ? reward : pre->r;
            } else {
                int pk = path[depth - 1 - i];
                p = pk >> 6; pa = pk & 63;
                pw = view(d, m, p);
                nv = pw.hdr->visits; V = pw.hdr->value; tp = pw.hdr->to_play;
                r = i == 0 ? reward : pw.crew[pa];
                cvs = pw.cvis.get(pa);
            }
        }
        int cnt = depth - base < LANES ? depth - base : LANES;
        lds_r[a] = r;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double myL = 0;
        int j = 0;
        for (; j + 4 <= cnt; j += 4) {
            double r0 = lds_r[j], r1 = lds_r[j + 1], r2 = lds_r[j + 2], r3 = lds_r[j + 3];
            if (a == j) myL = L;
            L = r0 + d.discount * L;
            if (a == j + 1) myL = L;
            L = r1 + d.discount * L;
            if (a == j + 2) myL = L;
            L = r2 + d.discount * L;
            if (a == j + 3) myL = L;
            L = r3 + d.discount * L;
        }
        for (; j < cnt; j++) {
            double rj = lds_r[j];
            if (a == j) myL = L;
            L = rj + d.discount * L;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double newV = (V * (double)nv + (tp == ntp ? myL : -myL)) / (double)(nv + 1);
        double below;  // the level below's new mean: lane a - 1 (a register move inside a row of 16 lanes)
        if (LANES <= 16 || cnt <= 16) below = dpp_mov<MZ_DPP_SHR(1)>(newV);
        else below = __shfl_up(newV, 1, LANES);
        double child_new = a == 0 ? carry : below;
        if (base + LANES < depth) carry = gshfl<LANES>(newV, cnt - 1);
        if (valid) {
            pw.hdr->visits = nv + 1;
            pw.hdr->value = newV;
            pw.cval[pa] = child_new;
            pw.cvis.set(pa, cvs + 1);
        }
    }
}

// mcts/mcts.py:22-31 + utils/game_utils.py:71-78: this lane's prepared root prior logit from the
// network's float32 logit (Dirichlet mix in float64 for PUCT, shift by the max, illegal -> dtype min).
template <int LANES, bool GUMBEL>
__device__ __forceinline__ double root_prior(float lg, bool on, bool lgl, double dirichlet_a, double frac,
                                             int a, int A, bool shift_and_mask)
{
    if constexpr (GUMBEL) {
        float mx = gmaxf<LANES>(on ? lg : -__builtin_huge_valf());
        return (double)(lgl ? lg - mx : -3.4028234663852886e38f);
    } else {
        float e = on ? np_expf(lg) : 0.0f;
        float es = gsum_np<LANES, float>(e, a, A);
        float p = __fdiv_rn(e, es);
        float keep = (float)(1.0 - frac);
        double noised = (double)(keep * p) + frac * (on ? dirichlet_a : 0.0);
        const double tiny = 2.2250738585072014e-308;
        double l = log(noised < tiny ? tiny : noised);  // np.maximum keeps a NaN (softmax of logits beyond the float32 exp range)
        double mx = gmax<LANES>(on ? l : -__builtin_huge_val());
        if (!shift_and_mask) return l;
        return lgl ? l - mx : -1.7976931348623157e308;
    }
}

}  // namespace mz
