// Root inference + root rows in one launch (k_begin_net) and the whole simulation loop with in-kernel
// leaf evaluation by an LDS-resident network (k_search_net).
#pragma once
#include "dense_net.h"

namespace mz {

// Root of a search evaluated on the device: initial_inference (network.py:86-103: representation,
// normalisation, prediction, support_to_scalar) + root prior preparation + the root rows.
struct RootNetDev {
    const int32_t *rep_meta, *pred_meta;
    const float *rep_data, *pred_data;
    int rep_n_meta, rep_n_data, pred_n_meta, pred_n_data;
    int rep_scratch, pred_scratch;
    int obs_off, raw_off, hid_off, pol_off, val_off, obs_size, support, ngroups;
};

// The root of tree t by its lane group: programs `rw/rm` (representation) and `pw/pm` (prediction) staged in
// LDS by the block, `sr`/`sp` this group's activation scratch for the two.
template <int LANES, int RL, bool GUMBEL>
__device__ __forceinline__ void begin_group(const Dev &d, const RootNetDev &rn, const float *rw, const int32_t *rm,
                                            const float *pw, const int32_t *pm, float *sr, float *sp, int t, int a,
                                            const float *obs, const uint8_t *legal, const int32_t *to_play,
                                            const double *dirichlet, double frac, const double *gumbel,
                                            int max_considered, float *out_logits, double *out_value)
{
    const int A = d.L.A, H = d.L.H;
    bool on = a < A;
    for (int i = a; i < rn.obs_size; i += LANES) sr[rn.obs_off + i] = obs[(long long)t * rn.obs_size + i];
    lds_fence();
    dense_run<LANES>(rm, rw, sr, a);
    minmax_to<LANES>(sr + rn.raw_off, H, rn.ngroups, a, d.hidden + (long long)t * d.L.N * H, sp + rn.hid_off);
    lds_fence();
    dense_run<LANES>(pm, pw, sp, a);
    double value = support_to_scalar<LANES>(sp + rn.val_off, rn.support, a);
    float lg = on ? sp[rn.pol_off + a] : 0.0f;
    long long k = (long long)t * A + a;
    bool lgl = on && legal[k] != 0;
    double di = (!GUMBEL && on) ? dirichlet[k] : 0.0;
    double pr = root_prior<RL, GUMBEL>(lg, on, lgl, di, frac, a, A, true);
    write_root<LANES, GUMBEL>(d, t, a, on, pr, value, to_play[t], lgl, (GUMBEL && on) ? gumbel[k] : 0.0,
                              max_considered);
    if (out_logits && on) out_logits[k] = lg;
    if (out_value && a == 0) out_value[t] = value;
}

template <int LANES, int RL, bool GUMBEL>
__global__ __launch_bounds__(MZ_BLOCK) void k_begin_net(Dev d, RootNetDev rn, const float *obs,
                                                         const uint8_t *legal, const int32_t *to_play,
                                                         const double *dirichlet, double frac,
                                                         const double *gumbel, int max_considered,
                                                         float *out_logits, double *out_value)
{
    extern __shared__ __align__(16) float lds[];
    float *rw = lds;
    int32_t *rm = reinterpret_cast<int32_t *>(rw + rn.rep_n_data);
    float *pw = reinterpret_cast<float *>(rm + rn.rep_n_meta);
    int32_t *pm = reinterpret_cast<int32_t *>(pw + rn.pred_n_data);
    float *scratch0 = reinterpret_cast<float *>(pm + rn.pred_n_meta);
    for (int i = threadIdx.x; i < rn.rep_n_data; i += MZ_BLOCK) rw[i] = rn.rep_data[i];
    for (int i = threadIdx.x; i < rn.rep_n_meta; i += MZ_BLOCK) rm[i] = rn.rep_meta[i];
    for (int i = threadIdx.x; i < rn.pred_n_data; i += MZ_BLOCK) pw[i] = rn.pred_data[i];
    for (int i = threadIdx.x; i < rn.pred_n_meta; i += MZ_BLOCK) pm[i] = rn.pred_meta[i];
    const int per_group = rn.rep_scratch + rn.pred_scratch;
    float *sr = scratch0 + (size_t)(threadIdx.x / LANES) * per_group, *sp = sr + rn.rep_scratch;
    for (int i = threadIdx.x % LANES; i < per_group; i += LANES) sr[i] = 0.0f;
    __syncthreads();
    int t, a;
    if (!group_ids<LANES>(d, &t, &a)) return;
    begin_group<LANES, RL, GUMBEL>(d, rn, rw, rm, pw, pm, sr, sp, t, a, obs, legal, to_play, dirichlet, frac, gumbel,
                                   max_considered, out_logits, out_value);
}

// TREE_LDS: each group's node records, hidden states and path live in LDS for the whole search (staged in
// at the start, written back at the end); `tree_lds_bytes` is the per-group size of that region.
template <int LANES, int RL, bool GUMBEL, bool TREE_LDS>
__global__ __launch_bounds__(MZ_BLOCK) void k_search_net(Dev d, NetDev net, int first_leaf, int n_sims,
                                                          const double *noise, int noise_max_calls,
                                                          int scratch_floats, int tree_lds_bytes)
{
    extern __shared__ __align__(16) float lds[];
    float *w = lds;                                                  // [n_data]
    int32_t *meta = reinterpret_cast<int32_t *>(lds + net.n_data);   // [n_meta]
    float *scratch0 = lds + net.n_data + net.n_meta;
    // the block may be launched with fewer than MZ_BLOCK threads (a multiple of LANES): small batches
    // then spread over more CUs
    const int n_threads = blockDim.x, groups_per_block = n_threads / LANES;
    for (int i = threadIdx.x; i < net.n_data; i += n_threads) w[i] = net.data[i];
    for (int i = threadIdx.x; i < net.n_meta; i += n_threads) meta[i] = net.meta[i];
    float *sc = scratch0 + (size_t)(threadIdx.x / LANES) * scratch_floats;
    for (int i = threadIdx.x % LANES; i < scratch_floats; i += LANES) sc[i] = 0.0f;
    __shared__ __align__(16) double s_bk[MZ_BLOCK];
    __shared__ __align__(16) double s_ct[GUMBEL ? 2 : MZ_CT_LDS];
    const double *ct = stage_c_table<GUMBEL>(d, s_ct);
    __shared__ __align__(16) double s_sq[GUMBEL ? 2 : MZ_SQ_LDS];
    const double *sqt = stage_sqrt_table<GUMBEL>(d, s_sq);
    __syncthreads();
    const int t = (int)blockIdx.x * groups_per_block + (int)threadIdx.x / LANES, a = threadIdx.x % LANES;
    if (t >= d.B) return;
    const int A = d.L.A, H = d.L.H;
    bool on = a < A;
    const double *nz = noise ? noise + (long long)t * noise_max_calls * A : nullptr;
    float *hid = d.hidden + (long long)t * d.L.N * H;
    float *const hid_global = hid;
    TreeMem m = tree_mem(d, t);
    const TreeMem m_global = m;
    if constexpr (TREE_LDS) {
        // region after the groups' activation scratch, 16-byte aligned: records | hidden | path
        size_t base = ((size_t)(net.n_data + net.n_meta) + (size_t)groups_per_block * scratch_floats) * 4;
        base = (base + 15) & ~(size_t)15;
        char *tl = reinterpret_cast<char *>(lds) + base + (size_t)(threadIdx.x / LANES) * tree_lds_bytes;
        const int rec_bytes = (d.L.N * d.L.rec + 15) & ~15, hid_bytes = (d.L.N * H * 4 + 15) & ~15;
        m.nodes = tl;
        hid = reinterpret_cast<float *>(tl + rec_bytes);
        m.path = reinterpret_cast<int32_t *>(tl + rec_bytes + hid_bytes);
        m.depth = m.path + ((d.L.S + 3) & ~3);
        const int32_t *src = reinterpret_cast<const int32_t *>(m_global.nodes);
        int32_t *dst = reinterpret_cast<int32_t *>(m.nodes);
        for (int i = a; i < first_leaf * (d.L.rec / 4); i += LANES) dst[i] = src[i];
        for (int i = a; i < first_leaf * H; i += LANES) hid[i] = hid_global[i];
        lds_fence();
        if (a == 0) {  // parents precede their children, so one forward sweep gives every depth
            m.depth[0] = 0;
            for (int n = 1; n < first_leaf; n++) m.depth[n] = m.depth[view(d, m, n).hdr->parent] + 1;
        }
        lds_fence();
    }
    constexpr bool WIDE = TREE_LDS && LANES == 64 && RL == 2;
    const bool wide = WIDE && d.L.N <= 2 * (64 / RL);
    double *bk = s_bk + (threadIdx.x / LANES) * LANES;
    DescendCtx dc = descend_ctx<GUMBEL>(d, t, a, on, ct, sqt);
    uint32_t calls = GUMBEL ? 0u : d.noise_calls[t];
    long long visited = 0;
    const int asz = net.io[8], sup = net.io[9], ngroups = net.io[10];
    const bool small_epilogue = ngroups == 1 && H <= 16 && sup <= 32;
    constexpr bool CAN_CACHE = LANES == 64;
    DenseRegs<2, 4> regs;
    bool cached = false;
    if constexpr (CAN_CACHE) {
        cached = dense_cacheable<LANES, 2, 4>(meta);
        if (cached) dense_cache_load<LANES, 2, 4>(meta, w, a, regs);
    }
#ifdef MZ_STAMPS
    long long st_sel = 0, st_net = 0, st_epi = 0, st_bk = 0, st_in = 0, st_mm = 0, st_t;
#define MZ_STAMP(acc) { long long n_ = clock64(); acc += n_ - st_t; st_t = n_; }
    st_t = clock64();
#else
#define MZ_STAMP(acc)
#endif
    for (int s = 0; s < n_sims; s++) {
        int leaf = first_leaf + s;
        int action, ntp, depth;
        int parent;
        if (wide) parent = descend_wide<RL, GUMBEL>(d, dc, m, a, leaf, nz, noise_max_calls, calls, &action, &ntp, &depth);
        else parent = descend<LANES, RL, GUMBEL>(d, dc, m, a, on, nz, noise_max_calls, calls, &action, &ntp, &depth);
        visited += depth + 1;
        MZ_STAMP(st_sel)
        // inputs: the PARENT's hidden state and the encoded action (mcts/mcts.py:240-243)
        const float *ph = hid + (long long)parent * H;
        for (int i = a; i < H; i += LANES) sc[net.io[0] + i] = ph[i];
        for (int i = a; i < asz; i += LANES) sc[net.io[1] + i] = i == action ? 1.0f : 0.0f;
        lds_fence();
        MZ_STAMP(st_in)
        if (CAN_CACHE && cached) dense_run_cached<2, 4>(regs, sc);
        else dense_run<LANES>(meta, w, sc, a);
        MZ_STAMP(st_net)
        double value, reward;
        if (LANES == 64 && small_epilogue) {
            epilogue_wave_small(sc + net.io[2], H, sc + net.io[5], sc + net.io[3], sup, a, hid + (long long)leaf * H,
                                &value, &reward);
            MZ_STAMP(st_mm)
        } else {
            minmax_to<LANES>(sc + net.io[2], H, ngroups, a, hid + (long long)leaf * H, nullptr);
            MZ_STAMP(st_mm)
            support_to_scalar2<LANES>(sc + net.io[5], sc + net.io[3], sup, a, &value, &reward);
        }
        float lg = on ? sc[net.io[4] + a] : 0.0f;
        MZ_STAMP(st_epi)
        expand_backup<LANES, GUMBEL>(d, m, a, on, parent, action, ntp, depth, leaf, lg, value, reward, bk);
        if constexpr (TREE_LDS) lds_fence();
        else __threadfence_block();
        MZ_STAMP(st_bk)
    }
    if (a == 0) {
        if constexpr (!GUMBEL) store_noise_calls(d, t, calls, noise_max_calls);
        d.visit_total[t] += visited;
    }
    if constexpr (TREE_LDS) {
        const int live = first_leaf + n_sims;
        const int32_t *src = reinterpret_cast<const int32_t *>(m.nodes);
        int32_t *dst = reinterpret_cast<int32_t *>(m_global.nodes);
        for (int i = a; i < live * (d.L.rec / 4); i += LANES) dst[i] = src[i];
        for (int i = a; i < live * H; i += LANES) hid_global[i] = hid[i];
    }
#ifdef MZ_STAMPS
    if (a == 0) {  // diagnostic build only: cycle shares per phase, parked in the path buffer (dead after the search)
        long long *o = reinterpret_cast<long long *>(d.path + (long long)t * d.L.S);
        o[0] = st_sel; o[1] = st_net; o[2] = st_epi; o[3] = st_bk; o[4] = st_in; o[5] = st_mm;
    }
#endif
}

}  // namespace mz
