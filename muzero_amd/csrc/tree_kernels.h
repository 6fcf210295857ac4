// Kernels that only touch trees: root rows, one selection / expansion+backup step per launch, the
// whole simulation loop with the synthetic hash network, the final action rule, root-prior
// preparation, the per-node entry points and the counters.
#pragma once
#include "tree_ops.h"

namespace mz {

#define MZ_BLOCK 256

#define MZ_CT_LDS 512  // exploration-constant entries staged in LDS (covers n_simulations <= 510)

// Copies the PUCT exploration-constant table into LDS (all threads of the block, before any exits).
template <bool GUMBEL>
__device__ __forceinline__ const double *stage_c_table(const Dev &d, double *lds_ct)
{
    if constexpr (GUMBEL) return nullptr;
    if (d.L.S + 2 > MZ_CT_LDS) return nullptr;
    for (int i = threadIdx.x; i < d.L.S + 2; i += blockDim.x) lds_ct[i] = d.c_table[i];
    __syncthreads();
    return lds_ct;
}

#define MZ_SQ_LDS (53 * 52)  // sqrt-table entries staged in LDS (covers n_simulations <= 51)

template <bool GUMBEL>
__device__ __forceinline__ const double *stage_sqrt_table(const Dev &d, double *lds_sq)
{
    if constexpr (GUMBEL) return nullptr;
    const int n = (d.L.S + 2) * (d.L.S + 1);
    if (!d.sqrt_table || n > MZ_SQ_LDS) return nullptr;
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds_sq[i] = d.sqrt_table[i];
    __syncthreads();
    return lds_sq;
}

template <int LANES>
__device__ __forceinline__ bool group_ids(const Dev &d, int *t, int *a)
{
    int gid = (int)((blockIdx.x * (unsigned)MZ_BLOCK + threadIdx.x) / LANES);
    *t = gid;
    *a = threadIdx.x % LANES;
    return gid < d.B;
}

// Root rows + per-tree search state (mcts/mcts.py:93-96 on top of the lazily initialised storage).
// `prior` is this lane's prepared root prior logit, `is_legal` this lane's legality.
template <int LANES, bool GUMBEL>
__device__ __forceinline__ void write_root(const Dev &d, int t, int a, bool on, double prior, double val,
                                           int to_play, bool is_legal, double gumbel_a, int max_considered)
{
    RowView w = view(d, tree_mem(d, t), 0);
    if (a == 0) {
        NodeHdr h;
        h.visits = 1; h.parent = -1; h.afp = -1; h.to_play = to_play; h.value = val; h.raw = val;
        *w.hdr = h;
        d.noise_calls[t] = 0;
        d.visit_total[t] = 0;
        d.sel_depth[t] = -1;
    }
    unsigned long long mask = __ballot(on && is_legal);
    int sh = (int)((threadIdx.x % 64) / LANES) * LANES;  // this group's lanes inside the 64-bit ballot
    if constexpr (LANES < 64) mask = (mask >> sh) & ((1ull << LANES) - 1ull);
    if (on) {
        w.cval[a] = 0.0; w.crew[a] = 0.0; w.cvis.set(a, 0); w.cidx.set(a, -1);
        if (GUMBEL) {
            reinterpret_cast<float *>(w.prior)[a] = (float)prior;
            d.gumbel[(long long)t * d.L.A + a] = gumbel_a;
        } else {
            reinterpret_cast<double *>(w.prior)[a] = prior;
        }
    }
    if (a == 0) {
        d.legal[t] = mask;
        int n_legal = __popcll(mask);
        d.m_considered[t] = max_considered < n_legal ? max_considered : n_legal;
    }
}

template <int LANES, int RL, bool GUMBEL>
__global__ __launch_bounds__(MZ_BLOCK) void k_reset(Dev d, const void *root_prior, const double *root_value,
                                                     const float *root_hidden, const int32_t *root_to_play,
                                                     const uint8_t *legal, const double *gumbel,
                                                     int max_considered)
{
    int t, a;
    if (!group_ids<LANES>(d, &t, &a)) return;
    const int A = d.L.A;
    bool on = a < A;
    long long k = (long long)t * A + a;
    double prior = 0.0, g = 0.0;
    if (on) {
        prior = GUMBEL ? (double)reinterpret_cast<const float *>(root_prior)[k]
                       : reinterpret_cast<const double *>(root_prior)[k];
        if (GUMBEL) g = gumbel[k];
    }
    write_root<LANES, GUMBEL>(d, t, a, on, prior, root_value[t], root_to_play[t], on && legal[k] != 0, g,
                              max_considered);
    if (root_hidden) {
        float *dst = d.hidden + (long long)t * d.L.N * d.L.H;
        for (int i = a; i < d.L.H; i += LANES) dst[i] = root_hidden[(long long)t * d.L.H + i];
    }
}

template <int LANES, int RL, bool GUMBEL>
__global__ __launch_bounds__(MZ_BLOCK) void k_select(Dev d, const double *noise, int noise_max_calls,
                                                      float *out_parent_hidden, int32_t *out_parent,
                                                      int32_t *out_action)
{
    __shared__ __align__(16) double s_ct[GUMBEL ? 2 : MZ_CT_LDS];
    const double *ct = stage_c_table<GUMBEL>(d, s_ct);
    __shared__ __align__(16) double s_sq[GUMBEL ? 2 : MZ_SQ_LDS];
    const double *sqt = stage_sqrt_table<GUMBEL>(d, s_sq);
    int t, a;
    if (!group_ids<LANES>(d, &t, &a)) return;
    bool on = a < d.L.A;
    const double *nz = noise ? noise + (long long)t * noise_max_calls * d.L.A : nullptr;
    int action, ntp, depth;
    DescendCtx dc = descend_ctx<GUMBEL>(d, t, a, on, ct, sqt);
    uint32_t calls = GUMBEL ? 0u : d.noise_calls[t];
    int parent = descend<LANES, RL, GUMBEL>(d, dc, tree_mem(d, t), a, on, nz, noise_max_calls, calls, &action, &ntp,
                                            &depth);
    if (a == 0) {
        if constexpr (!GUMBEL) store_noise_calls(d, t, calls, noise_max_calls);
        d.sel_parent[t] = parent; d.sel_action[t] = action; d.sel_ntp[t] = ntp; d.sel_depth[t] = depth;
        if (out_parent) out_parent[t] = parent;
        if (out_action) out_action[t] = action;
    }
    if (out_parent_hidden) {
        const float *src = d.hidden + ((long long)t * d.L.N + parent) * d.L.H;
        for (int i = a; i < d.L.H; i += LANES) out_parent_hidden[(long long)t * d.L.H + i] = src[i];
    }
}

template <int LANES, int RL, bool GUMBEL>
__global__ __launch_bounds__(MZ_BLOCK) void k_expand_backup(Dev d, int leaf, const float *logits,
                                                             const double *value, const double *reward,
                                                             const float *next_hidden)
{
    __shared__ __align__(16) double s_bk[MZ_BLOCK];
    int t, a;
    if (!group_ids<LANES>(d, &t, &a)) return;
    bool on = a < d.L.A;
    int parent = d.sel_parent[t], action = d.sel_action[t], ntp = d.sel_ntp[t], depth = d.sel_depth[t];
    if (depth < 1 || depth > d.L.S || parent < 0 || parent >= leaf) {  // select did not run for this leaf
        if (a == 0) atomicExch(d.err, 1);
        return;
    }
    float lg = on ? logits[(long long)t * d.L.A + a] : 0.0f;
    expand_backup<LANES, GUMBEL>(d, tree_mem(d, t), a, on, parent, action, ntp, depth, leaf, lg, value[t], reward[t],
                                 s_bk + (threadIdx.x / LANES) * LANES);
    if (a == 0) d.visit_total[t] += depth + 1;
    if (next_hidden) {
        float *dst = d.hidden + ((long long)t * d.L.N + leaf) * d.L.H;
        for (int i = a; i < d.L.H; i += LANES) dst[i] = next_hidden[(long long)t * d.L.H + i];
    }
    if (a == 0) d.sel_depth[t] = -1;
}

// k_expand_backup for the pending selection followed at once by the next simulation's k_select: the
// records the backup just wrote are still in the near caches when the descent re-reads them, and the
// step loop needs one launch less per simulation.
template <int LANES, int RL, bool GUMBEL>
__global__ __launch_bounds__(MZ_BLOCK) void k_backup_select(Dev d, int leaf, const float *logits,
                                                             const double *value, const double *reward,
                                                             const double *noise, int noise_max_calls)
{
    __shared__ __align__(16) double s_bk[MZ_BLOCK];
    __shared__ __align__(16) double s_ct[GUMBEL ? 2 : MZ_CT_LDS];
    const double *ct = stage_c_table<GUMBEL>(d, s_ct);
    __shared__ __align__(16) double s_sq[GUMBEL ? 2 : MZ_SQ_LDS];
    const double *sqt = stage_sqrt_table<GUMBEL>(d, s_sq);
    int t, a;
    if (!group_ids<LANES>(d, &t, &a)) return;
    bool on = a < d.L.A;
    int parent = d.sel_parent[t], action = d.sel_action[t], ntp = d.sel_ntp[t], depth = d.sel_depth[t];
    if (depth < 1 || depth > d.L.S || parent < 0 || parent >= leaf) {
        if (a == 0) atomicExch(d.err, 1);
        return;
    }
    const TreeMem m = tree_mem(d, t);
    float lg = on ? logits[(long long)t * d.L.A + a] : 0.0f;
    expand_backup<LANES, GUMBEL>(d, m, a, on, parent, action, ntp, depth, leaf, lg, value[t], reward[t],
                                 s_bk + (threadIdx.x / LANES) * LANES);
    if (a == 0) d.visit_total[t] += depth + 1;
    __threadfence_block();
    const double *nz = noise ? noise + (long long)t * noise_max_calls * d.L.A : nullptr;
    DescendCtx dc = descend_ctx<GUMBEL>(d, t, a, on, ct, sqt);
    uint32_t calls = GUMBEL ? 0u : d.noise_calls[t];
    parent = descend<LANES, RL, GUMBEL>(d, dc, m, a, on, nz, noise_max_calls, calls, &action, &ntp, &depth);
    if (a == 0) {
        if constexpr (!GUMBEL) store_noise_calls(d, t, calls, noise_max_calls);
        d.sel_parent[t] = parent; d.sel_action[t] = action; d.sel_ntp[t] = ntp; d.sel_depth[t] = depth;
    }
}

// Whole search in one launch with the synthetic hash network as the leaf evaluator.
template <int LANES, int RL, bool GUMBEL>
__global__ __launch_bounds__(MZ_BLOCK) void k_search_hash(Dev d, HashNet hn, int first_leaf, int n_sims,
                                                           const double *noise, int noise_max_calls)
{
    __shared__ __align__(16) double s_bk[MZ_BLOCK];
    __shared__ __align__(16) double s_ct[GUMBEL ? 2 : MZ_CT_LDS];
    const double *ct = stage_c_table<GUMBEL>(d, s_ct);
    __shared__ __align__(16) double s_sq[GUMBEL ? 2 : MZ_SQ_LDS];
    const double *sqt = stage_sqrt_table<GUMBEL>(d, s_sq);
    int t, a;
    if (!group_ids<LANES>(d, &t, &a)) return;
    const int A = d.L.A, H = d.L.H;
    bool on = a < A;
    const double *nz = noise ? noise + (long long)t * noise_max_calls * A : nullptr;
    float *hid = d.hidden + (long long)t * d.L.N * H;
    double *bk = s_bk + (threadIdx.x / LANES) * LANES;
    DescendCtx dc = descend_ctx<GUMBEL>(d, t, a, on, ct, sqt);
    const TreeMem m = tree_mem(d, t);
    uint32_t calls = GUMBEL ? 0u : d.noise_calls[t];
    long long visited = 0;
    for (int s = 0; s < n_sims; s++) {
        int leaf = first_leaf + s;
        int action, ntp, depth;
        int parent = descend<LANES, RL, GUMBEL>(d, dc, m, a, on, nz, noise_max_calls, calls, &action, &ntp, &depth);
        visited += depth + 1;
        const float *ph = hid + (long long)parent * H;
        uint32_t pcode = (uint32_t)ph[0] | ((uint32_t)ph[1] << 16);
        uint32_t ccode = hash2(pcode, 300 + action);
        uint32_t src = hn.parent_predicts ? pcode : ccode;
        float lg = on ? hash_logit(src, a, hn.logit_shift) : 0.0f;
        expand_backup<LANES, GUMBEL>(d, m, a, on, parent, action, ntp, depth, leaf, lg, hash_value(src),
                                     hash_reward(ccode, hn.reward_on), bk);
        float *dst = hid + (long long)leaf * H;
        for (int i = a; i < H; i += LANES) dst[i] = hash_hidden(ccode, i);
        __threadfence_block();  // this group's stores are read by its other lanes in the next descent
    }
    if (a == 0) {
        if constexpr (!GUMBEL) store_noise_calls(d, t, calls, noise_max_calls);
        d.visit_total[t] += visited;
    }
}



// mcts/action_selection.py:52-70 (PUCT) / :121-138 (Gumbel) + mcts/tree.py:60-68: the action to play,
// this lane's policy entry and q entry, and the root value returned by MCTS.search (mcts/mcts.py:252).
template <int RL, bool GUMBEL>
__device__ __forceinline__ int to_act_rule(const Dev &d, const TreeMem &m, int t, int a, bool on, double temperature,
                                           double uu, double *prob_out, double *q_out, double *root_value_out)
{
    const int A = d.L.A;
    const double inf = __builtin_huge_val();
    NodeStats s = load_node<GUMBEL>(d, m, 0, a, on);
    bool legal = (d.legal[t] >> a) & 1ull;
    int action;
    double prob, qout;
    if constexpr (!GUMBEL) {
        int total = gsumi<RL>(on ? s.cv : 0);
        int den = total > 1 ? total : 1;
        double vp = total > 0 ? (double)s.cv / (double)den : 1.0 / (double)A;
        const double tiny = 2.2250738585072014e-308;
        double lg = log(vp > tiny ? vp : tiny);
        double mx = gmax<RL>(on ? lg : -inf);
        double tt = temperature > 1e-27 ? temperature : 1e-27;
        lg = (lg - mx) / tt;
        double e = on ? exp(lg) : 0.0;
        double es = gsum_np<RL, double>(e, a, A);
        prob = e / es;
        // legacy np.random.choice: cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, side='right')
        double acc = 0.0, mine = 0.0;
        for (int i = 0; i < A; i++) { acc += __shfl(prob, i, RL); if (a == i) mine = acc; }
        mine = mine / acc;
        int idx = gsumi<RL>((on && mine <= uu) ? 1 : 0);
        action = idx < A ? idx : A - 1;
        qout = s.r + d.discount * s.v;
    } else {
        int vsum;
        double cq = qt_completed_mix<RL>(d, s, a, on, &vsum);
        int considered = gmaxi<RL>(on ? s.cv : -2147483647);
        double sc = gumbel_score<RL>(s, on, on ? d.gumbel[(long long)t * A + a] : 0.0, cq, considered);
        sc = sc != sc ? inf : sc;   // np.argmax: the first NaN wins
        if (!on || !legal) sc = -inf;
        action = gargmax<RL>(sc, on ? a : 0x7fffffff);
        double x = s.prior + cq;
        double mx = gmax<RL>(on ? x : -inf);
        x = legal ? x - mx : -1.7976931348623157e308;
        double e = on ? exp(x) : 0.0;
        double es = gsum_np<RL, double>(e, a, A);
        prob = e / es;
        qout = cq;
    }
    *prob_out = prob; *q_out = qout; *root_value_out = s.nval;
    return action;
}

template <int LANES, int RL, bool GUMBEL>
__global__ __launch_bounds__(MZ_BLOCK) void k_to_act(Dev d, double temperature, const double *u,
                                                      int32_t *out_action, double *out_probs,
                                                      double *out_root_value, double *out_q)
{
    int t, a;
    if (!group_ids<LANES>(d, &t, &a)) return;
    const int A = d.L.A;
    bool on = a < A;
    double prob, qout, rv;
    int action = to_act_rule<RL, GUMBEL>(d, tree_mem(d, t), t, a, on, temperature, u ? u[t] : 0.0, &prob, &qout, &rv);
    if (on) {
        out_probs[(long long)t * A + a] = prob;
        if (out_q) out_q[(long long)t * A + a] = qout;
    }
    if (a == 0) { out_action[t] = action; out_root_value[t] = rv; }
}


// mcts/mcts.py:22-31 + utils/game_utils.py:71-78 for B roots (root_prior in tree_ops.h).
template <int LANES, int RL, bool GUMBEL>
__global__ __launch_bounds__(MZ_BLOCK) void k_root_prepare(int B, int A, const float *logits,
                                                            const double *dirichlet, double frac,
                                                            const uint8_t *legal, int shift_and_mask,
                                                            void *out)
{
    int gid = (int)((blockIdx.x * (unsigned)MZ_BLOCK + threadIdx.x) / LANES), a = threadIdx.x % LANES;
    if (gid >= B) return;
    bool on = a < A;
    long long k = (long long)gid * A + a;
    float lg = on ? logits[k] : 0.0f;
    bool lgl = on && (legal ? legal[k] != 0 : true);
    double di = (!GUMBEL && on) ? dirichlet[k] : 0.0;
    double pr = root_prior<RL, GUMBEL>(lg, on, lgl, di, frac, a, A, shift_and_mask != 0);
    if (on) {
        if (GUMBEL) reinterpret_cast<float *>(out)[k] = (float)pr;
        else reinterpret_cast<double *>(out)[k] = pr;
    }
}

// One rule at one node, for the module-level functions of the drop-in package.
template <int LANES, int RL, bool GUMBEL>
__global__ __launch_bounds__(64) void k_node_op(Dev d, int t, int node, int depth, int which,
                                                 const double *uniform, double *out_vec, int32_t *out_action)
{
    int a = threadIdx.x % LANES;
    if (threadIdx.x >= LANES) return;
    bool on = a < d.L.A;
    if (which == 3) {  // selection
        int child;
        double u = (uniform && on) ? uniform[a] : 0.0;
        DescendCtx dc = descend_ctx<GUMBEL>(d, t, a, on, nullptr);
        int act = select_at<RL, GUMBEL>(d, dc, tree_mem(d, t), node, depth, a, on, u, &child);
        if (a == 0) *out_action = act;
        return;
    }
    NodeStats s = load_node<GUMBEL>(d, tree_mem(d, t), node, a, on);
    double r;
    if (which == 0) {
        r = qt_parent_siblings<RL>(d, s, on);
    } else {
        int vsum;
        if constexpr (GUMBEL) {
            if (which == 1) {
                r = qt_completed_mix<RL>(d, s, a, on, &vsum);
            } else {
                vsum = gsumi<RL>(on ? s.cv : 0);
                r = mixed_value<RL>(d, s, a, on, s.r + d.discount * s.v, vsum);
            }
        } else {
            r = 0.0;
        }
    }
    if (on) out_vec[a] = r;
}

// MCTS.backward (mcts/mcts.py:143-164) from an arbitrary node, walking the parent links; one lane.
static __global__ void k_node_backward(Dev d, int t, int leaf, int to_play)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const TreeMem m = tree_mem(d, t);
    double leaf_value = view(d, m, leaf).hdr->value;
    for (int guard = 0; leaf != 0 && guard <= d.L.S; guard++) {
        RowView lf = view(d, m, leaf);
        int p = lf.hdr->parent, a = lf.hdr->afp;
        if (p < 0 || p >= d.L.N || a < 0 || a >= d.L.A) { atomicExch(d.err, 1); return; }
        RowView pw = view(d, m, p);
        double reward = pw.crew[a];
        int pv = pw.hdr->visits;
        double sv = pw.hdr->to_play == to_play ? leaf_value : -leaf_value;
        pw.hdr->value = (pw.hdr->value * (double)pv + sv) / (double)(pv + 1);
        pw.hdr->visits = pv + 1;
        pw.cval[a] = lf.hdr->value;
        pw.cvis.set(a, pw.cvis.get(a) + 1);
        leaf_value = reward + d.discount * leaf_value;
        leaf = p;
    }
}

static __global__ void k_stats(Dev d, long long *out_node_visits, int32_t *out_noise_calls)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= d.B) return;
    if (out_node_visits) out_node_visits[t] = d.visit_total[t];
    if (out_noise_calls) out_noise_calls[t] = (int32_t)d.noise_calls[t];
}

}  // namespace mz
