// Kernels and C ABI of the batched search engine (gfx950).  See include/muzero_amd.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/muzero_amd.h"
#include "tree_ops.h"
#include "targets.h"

using namespace mz;

// ============================================================================ kernels

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
        w.cval[a] = 0.0; w.crew[a] = 0.0; w.cvis[a] = 0; w.cidx[a] = -1;
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
        if constexpr (!GUMBEL) d.noise_calls[t] = calls;
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
        if constexpr (!GUMBEL) d.noise_calls[t] = calls;
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
        if constexpr (!GUMBEL) d.noise_calls[t] = calls;
        d.visit_total[t] += visited;
    }
}


// ---------------------------------------------------------------------------- fused network search
// Leaf evaluation by a lowered network ("dense program", layout in muzero_amd/packing.py) whose
// weights fit in LDS: the block stages weights + tables once, every lane group owns an activation
// scratch in LDS, and a level's output rows are spread over the group's lanes.
struct NetDev {
    const int32_t *meta;
    const float *data;
    int n_meta, n_data;
    int io[12];  // in_hidden, in_action, out_next, out_reward, out_policy, out_value, H, A, action_size, support, norm_groups, obs
};

// utils/network_utils.py:75-96 over `n` logits held in LDS: float32 softmax, expectation over the
// integer support in float64, inverse of h(x) = sign(x)(sqrt(|x|+1)-1) + 0.001x.  Two vectors
// (value and reward logits) go through together so their dependent chains overlap.
__device__ __forceinline__ double inv_value_transform(double x)
{
    const double eps = 0.001;
    double sg = x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0);
    double t = (sqrt(1 + 4 * eps * (fabs(x) + 1 + eps)) - 1) / (2 * eps);
    return sg * (t * t - 1);
}

template <int LANES>
__device__ __forceinline__ void support_to_scalar2(const float *lgA, const float *lgB, int n, int a,
                                                   double *outA, double *outB)
{
    const int L = (n - 1) / 2;
    if constexpr (LANES == 64) {
        if (n <= 32) {  // one vector per half of the wave: a single set of half-wave reductions serves both
            const bool up = a >= 32;
            const int i = a & 31;
            const float ninf = -__builtin_huge_valf();
            float v = i < n ? (up ? lgB : lgA)[i] : ninf;
            float m = half_reduce(v, ninf, up, [](float x, float y) { return y > x ? y : x; });
            float e = i < n ? expf(v - m) : 0.0f;
            float sum = half_reduce(e, 0.0f, up, [](float x, float y) { return x + y; });
            double x = i < n ? (double)(e / sum) * (double)(i - L) : 0.0;
            x = half_reduce(x, 0.0, up, [](double p, double q) { return p + q; });
            double y = inv_value_transform(x);
            *outA = wave_lane(y, 0);
            *outB = wave_lane(y, 32);
            return;
        }
    }
    int W = 2;
    while (W < n && W < LANES) W <<= 1;  // lanes that hold elements
    const bool single = n <= LANES;      // one element per lane: keep it in registers
    float mA = -__builtin_huge_valf(), mB = -__builtin_huge_valf();
    float vA = 0, vB = 0;
    if (single) {
        if (a < n) { vA = lgA[a]; vB = lgB[a]; mA = vA; mB = vB; }
    } else {
        for (int i = a; i < n; i += LANES) { mA = lgA[i] > mA ? lgA[i] : mA; mB = lgB[i] > mB ? lgB[i] : mB; }
    }
    auto fmaxl = [](float x, float y) { return y > x ? y : x; };
    auto faddl = [](float x, float y) { return x + y; };
    auto daddl = [](double x, double y) { return x + y; };
    if constexpr (LANES == 64) { mA = wave_reduce(mA, -__builtin_huge_valf(), fmaxl); mB = wave_reduce(mB, -__builtin_huge_valf(), fmaxl); }
    else { mA = gmaxf<LANES>(mA, W); mB = gmaxf<LANES>(mB, W); }
    float sA = 0.0f, sB = 0.0f, eA = 0.0f, eB = 0.0f;
    if (single) {
        if (a < n) { eA = expf(vA - mA); eB = expf(vB - mB); }
        sA = eA; sB = eB;
    } else {
        for (int i = a; i < n; i += LANES) { sA += expf(lgA[i] - mA); sB += expf(lgB[i] - mB); }
    }
    if constexpr (LANES == 64) { sA = wave_reduce(sA, 0.0f, faddl); sB = wave_reduce(sB, 0.0f, faddl); }
    else { sA = gsumf_any<LANES>(sA, W); sB = gsumf_any<LANES>(sB, W); }
    double xA = 0.0, xB = 0.0;
    if (single) {
        if (a < n) { xA = (double)(eA / sA) * (double)(a - L); xB = (double)(eB / sB) * (double)(a - L); }
    } else {
        for (int i = a; i < n; i += LANES) {
            xA += (double)(expf(lgA[i] - mA) / sA) * (double)(i - L);
            xB += (double)(expf(lgB[i] - mB) / sB) * (double)(i - L);
        }
    }
    if constexpr (LANES == 64) { xA = wave_reduce(xA, 0.0, daddl); xB = wave_reduce(xB, 0.0, daddl); }
    else { xA = gsumd_any<LANES>(xA, W); xB = gsumd_any<LANES>(xB, W); }
    if constexpr (LANES == 64) {
        double y = inv_value_transform(a == 1 ? xB : xA);
        *outA = wave_lane(y, 0);
        *outB = wave_lane(y, 1);
    } else if constexpr (LANES >= 2) {
        // one pass of the (long, float64) inverse transform: lane 0 takes the first vector, lane 1 the second
        double y = inv_value_transform(a == 1 ? xB : xA);
        *outA = __shfl(y, 0, LANES);
        *outB = __shfl(y, 1, LANES);
    } else {
        *outA = inv_value_transform(xA);
        *outB = inv_value_transform(xB);
    }
}

template <int LANES>
__device__ __forceinline__ double support_to_scalar(const float *lg, int n, int a)
{
    double x, y;
    support_to_scalar2<LANES>(lg, lg, n, a, &x, &y);
    return x;
}

__device__ __forceinline__ void lds_fence()
{
    // LDS operations of one wave complete in order; this only stops the compiler from moving them.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One evaluation of a dense program by a lane group: rows of a level are spread over the lanes.
template <int LANES>
__device__ __forceinline__ void dense_run(const int32_t *meta, const float *w, float *sc, int a)
{
    const int n_levels = meta[0];
    const int32_t *lv = meta + meta[5], *rows = meta + meta[6];
    const float *bias = w + meta[4];
    for (int l = 0; l < n_levels; l++) {
        const int n_rows = lv[4 * l], kmax = lv[4 * l + 1], row0 = lv[4 * l + 3];
        const float *WT = w + lv[4 * l + 2];
        const float4 *W4 = reinterpret_cast<const float4 *>(WT);
        const int k4n = kmax >> 2;
        for (int r = a; r < n_rows; r += LANES) {
            const int4 ds = *reinterpret_cast<const int4 *>(rows + 4 * (row0 + r));
            const float4 *x4 = reinterpret_cast<const float4 *>(sc + ds.x);
            float acc = bias[row0 + r];
#pragma unroll 4
            for (int k = 0; k < k4n; k++) {
                float4 wv = W4[k * n_rows + r], xv = x4[k];
                acc = fmaf(wv.x, xv.x, acc);
                acc = fmaf(wv.y, xv.y, acc);
                acc = fmaf(wv.z, xv.z, acc);
                acc = fmaf(wv.w, xv.w, acc);
            }
            if (ds.w >= 0) acc += sc[ds.w];
            if (ds.z == 1) acc = acc > 0.0f ? acc : 0.0f;
            else if (ds.z == 2) acc = acc > 0.0f ? acc : expm1f(acc);
            sc[ds.y] = acc;
        }
        lds_fence();
    }
}

// A program of at most MAXL levels whose levels have at most one row per lane and at most 4 * MAXK4
// inputs: each lane keeps its row's descriptor, bias and weights in registers across evaluations, so an
// evaluation only reads the inputs from LDS.  Same operation order as dense_run.
template <int MAXL, int MAXK4>
struct DenseRegs {
    int4 ds[MAXL];
    float bias[MAXL];
    float4 w[MAXL][MAXK4];
    int k4n[MAXL];
    int n_levels;
};

template <int LANES, int MAXL, int MAXK4>
__device__ __forceinline__ bool dense_cacheable(const int32_t *meta)
{
    if (meta[0] > MAXL) return false;
    const int32_t *lv = meta + meta[5];
    for (int l = 0; l < meta[0]; l++)
        if (lv[4 * l] > LANES || lv[4 * l + 1] > 4 * MAXK4) return false;
    return true;
}

template <int LANES, int MAXL, int MAXK4>
__device__ __forceinline__ void dense_cache_load(const int32_t *meta, const float *w, int a, DenseRegs<MAXL, MAXK4> &c)
{
    const int32_t *lv = meta + meta[5], *rows = meta + meta[6];
    const float *bias = w + meta[4];
    c.n_levels = meta[0];
#pragma unroll
    for (int l = 0; l < MAXL; l++) {
        c.ds[l] = make_int4(0, -1, 0, -1);
        c.bias[l] = 0.0f;
        c.k4n[l] = 0;
#pragma unroll
        for (int k = 0; k < MAXK4; k++) c.w[l][k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l < c.n_levels) {
            const int n_rows = lv[4 * l], row0 = lv[4 * l + 3];
            const float4 *W4 = reinterpret_cast<const float4 *>(w + lv[4 * l + 2]);
            c.k4n[l] = lv[4 * l + 1] >> 2;
            if (a < n_rows) {
                c.ds[l] = *reinterpret_cast<const int4 *>(rows + 4 * (row0 + a));
                c.bias[l] = bias[row0 + a];
#pragma unroll
                for (int k = 0; k < MAXK4; k++)
                    if (k < c.k4n[l]) c.w[l][k] = W4[k * n_rows + a];
            }
        }
    }
}

template <int MAXL, int MAXK4>
__device__ __forceinline__ void dense_run_cached(const DenseRegs<MAXL, MAXK4> &c, float *sc)
{
#pragma unroll
    for (int l = 0; l < MAXL; l++) {
        if (l < c.n_levels) {
            const int4 ds = c.ds[l];
            if (ds.y >= 0) {
                const float4 *x4 = reinterpret_cast<const float4 *>(sc + ds.x);
                float acc = c.bias[l];
#pragma unroll
                for (int k = 0; k < MAXK4; k++) {
                    if (k < c.k4n[l]) {
                        float4 wv = c.w[l][k], xv = x4[k];
                        acc = fmaf(wv.x, xv.x, acc);
                        acc = fmaf(wv.y, xv.y, acc);
                        acc = fmaf(wv.z, xv.z, acc);
                        acc = fmaf(wv.w, xv.w, acc);
                    }
                }
                if (ds.w >= 0) acc += sc[ds.w];
                if (ds.z == 1) acc = acc > 0.0f ? acc : 0.0f;
                else if (ds.z == 2) acc = acc > 0.0f ? acc : expm1f(acc);
                sc[ds.y] = acc;
            }
            lds_fence();
        }
    }
}

// utils/network_utils.py:132-158: per-group min-max scaling of `raw` (LDS) into `dst` (and `dst2` if given).
template <int LANES>
__device__ __forceinline__ void minmax_to(const float *raw, int H, int ngroups, int a, float *dst, float *dst2)
{
    const int per = H / ngroups;
    for (int g = 0; g < ngroups; g++) {
        const float *r = raw + g * per;
        float mn = __builtin_huge_valf(), mx = -__builtin_huge_valf();
        for (int i = a; i < per; i += LANES) { float v = r[i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        int W = 2;
        while (W < per && W < LANES) W <<= 1;
        if constexpr (LANES == 64) {
            if (W <= 16) {  // the values sit in one row of 16 lanes
                mn = gminf<16>(mn, W); mx = gmaxf<16>(mx, W);
            } else {
                mn = wave_reduce(mn, __builtin_huge_valf(), [](float x, float y) { return y < x ? y : x; });
                mx = wave_reduce(mx, -__builtin_huge_valf(), [](float x, float y) { return y > x ? y : x; });
            }
        } else {
            mn = gminf<LANES>(mn, W); mx = gmaxf<LANES>(mx, W);
        }
        float den = (mx - mn) + 1e-5f;
        for (int i = a; i < per; i += LANES) {
            float v = (r[i] - mn) / den;
            dst[g * per + i] = v;
            if (dst2) dst2[g * per + i] = v;
        }
    }
}

#include "net_mfma.h"  // matrix-core leaf evaluation; uses the epilogue helpers above

// Root of a search evaluated on the device: initial_inference (network.py:86-103: representation,
// normalisation, prediction, support_to_scalar) + root prior preparation + the root rows.
struct RootNetDev {
    const int32_t *rep_meta, *pred_meta;
    const float *rep_data, *pred_data;
    int rep_n_meta, rep_n_data, pred_n_meta, pred_n_data;
    int rep_scratch, pred_scratch;
    int obs_off, raw_off, hid_off, pol_off, val_off, obs_size, support, ngroups;
};

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
    for (int i = threadIdx.x; i < net.n_data; i += MZ_BLOCK) w[i] = net.data[i];
    for (int i = threadIdx.x; i < net.n_meta; i += MZ_BLOCK) meta[i] = net.meta[i];
    float *sc = scratch0 + (size_t)(threadIdx.x / LANES) * scratch_floats;
    for (int i = threadIdx.x % LANES; i < scratch_floats; i += LANES) sc[i] = 0.0f;
    __shared__ __align__(16) double s_bk[MZ_BLOCK];
    __shared__ __align__(16) double s_ct[GUMBEL ? 2 : MZ_CT_LDS];
    const double *ct = stage_c_table<GUMBEL>(d, s_ct);
    __shared__ __align__(16) double s_sq[GUMBEL ? 2 : MZ_SQ_LDS];
    const double *sqt = stage_sqrt_table<GUMBEL>(d, s_sq);
    __syncthreads();
    int t, a;
    if (!group_ids<LANES>(d, &t, &a)) return;
    const int A = d.L.A, H = d.L.H;
    bool on = a < A;
    const double *nz = noise ? noise + (long long)t * noise_max_calls * A : nullptr;
    float *hid = d.hidden + (long long)t * d.L.N * H;
    float *const hid_global = hid;
    TreeMem m = tree_mem(d, t);
    const TreeMem m_global = m;
    if constexpr (TREE_LDS) {
        // region after the groups' activation scratch, 16-byte aligned: records | hidden | path
        size_t base = ((size_t)(net.n_data + net.n_meta) + (size_t)(MZ_BLOCK / LANES) * scratch_floats) * 4;
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
        minmax_to<LANES>(sc + net.io[2], H, ngroups, a, hid + (long long)leaf * H, nullptr);
        MZ_STAMP(st_mm)
        double value, reward;
        support_to_scalar2<LANES>(sc + net.io[5], sc + net.io[3], sup, a, &value, &reward);
        float lg = on ? sc[net.io[4] + a] : 0.0f;
        MZ_STAMP(st_epi)
        expand_backup<LANES, GUMBEL>(d, m, a, on, parent, action, ntp, depth, leaf, lg, value, reward, bk);
        if constexpr (TREE_LDS) lds_fence();
        else __threadfence_block();
        MZ_STAMP(st_bk)
    }
    if (a == 0) {
        if constexpr (!GUMBEL) d.noise_calls[t] = calls;
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

// mcts/action_selection.py:52-70 (PUCT) / :121-138 (Gumbel) + mcts/tree.py:60-68: the action to play,
// this lane's policy entry and q entry, and the root value returned by MCTS.search (mcts/mcts.py:252).
template <int RL, bool GUMBEL>
__device__ __forceinline__ int to_act_rule(const Dev &d, int t, int a, bool on, double temperature, double uu,
                                           double *prob_out, double *q_out, double *root_value_out)
{
    const int A = d.L.A;
    const double inf = __builtin_huge_val();
    NodeStats s = load_node<GUMBEL>(d, tree_mem(d, t), 0, a, on);
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
    int action = to_act_rule<RL, GUMBEL>(d, t, a, on, temperature, u ? u[t] : 0.0, &prob, &qout, &rv);
    if (on) {
        out_probs[(long long)t * A + a] = prob;
        if (out_q) out_q[(long long)t * A + a] = qout;
    }
    if (a == 0) { out_action[t] = action; out_root_value[t] = rv; }
}

// ---------------------------------------------------------------------------- self-play step
// Final action rule fused with the environment step of self_play.py:92-113 for every env of the batch:
// act, game.step(action), hand the trajectory record of this move to the caller, restart finished
// games.  Environment state is caller-owned device memory (mz_env_state in the header).
struct EnvDev {
    int kind, max_moves, stale, obs_size;
    void *state;
    int32_t *mover, *to_play;
    uint8_t *legal;
    float *cur_obs, *search_obs;
    int32_t *moves;
    const double *reset_state;
};

// games/tictactoe.py:14-81 + games/game.py:56-74.  Returns terminated; winner: -2 = none yet.
__device__ __forceinline__ bool tictactoe_step(signed char *b, int *mover, int action, float *reward)
{
    b[action] = (signed char)*mover;  // no occupancy check, like the reference
    *mover = -*mover;
    int winner = -2;
    bool term = false;
    const int lines[8][3] = {{0, 3, 6}, {1, 4, 7}, {2, 5, 8}, {0, 1, 2}, {3, 4, 5}, {6, 7, 8}, {0, 4, 8}, {2, 4, 6}};
    for (int i = 0; i < 8; i++) {  // columns, rows, diagonal, anti-diagonal: first complete line wins
        int sline = b[lines[i][0]] + b[lines[i][1]] + b[lines[i][2]];
        if (sline == 3 || sline == -3) { term = true; winner = sline / 3; break; }
    }
    int filled = 0;
    for (int i = 0; i < 9; i++) filled += b[i] != 0;
    if (filled == 9) { term = true; winner = 0; }  // a board-filling win is scored as a tie
    *reward = (winner != -2 && winner != 0) ? 1.0f : 0.0f;
    return term;
}

__device__ __forceinline__ void tictactoe_observe(const signed char *b, int mover, float *obs, uint8_t *legal)
{
    for (int i = 0; i < 9; i++) {
        obs[i] = b[i] == -1 ? 1.0f : 0.0f;
        obs[9 + i] = b[i] == 1 ? 1.0f : 0.0f;
        obs[18 + i] = (float)mover;
        legal[i] = b[i] == 0;
    }
}

// CartPole-v1 per the public gymnasium specification (games/cartpole.py:9-48 delegates to it): Euler,
// tau 0.02, float64 state, float32 observation, reward 1 per step, terminated outside +-2.4 / +-12 degrees.
__device__ __forceinline__ bool cartpole_step(double *s, int action)
{
    const double gravity = 9.8, masspole = 0.1, total_mass = 1.1, length = 0.5;
    const double polemass_length = masspole * length, force_mag = 10.0, tau = 0.02;
    const double theta_thr = 12 * 2 * 3.141592653589793 / 360, x_thr = 2.4;
    double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
    double force = action == 1 ? force_mag : -force_mag;
    double costheta = cos(theta), sintheta = sin(theta);
    double temp = (force + polemass_length * (theta_dot * theta_dot) * sintheta) / total_mass;
    double thetaacc = (gravity * sintheta - costheta * temp) /
                      (length * (4.0 / 3.0 - masspole * (costheta * costheta) / total_mass));
    double xacc = temp - polemass_length * thetaacc * costheta / total_mass;
    x = x + tau * x_dot; x_dot = x_dot + tau * xacc;
    theta = theta + tau * theta_dot; theta_dot = theta_dot + tau * thetaacc;
    s[0] = x; s[1] = x_dot; s[2] = theta; s[3] = theta_dot;
    return x < -x_thr || x > x_thr || theta < -theta_thr || theta > theta_thr;
}

template <int LANES, int RL, bool GUMBEL>
__global__ __launch_bounds__(MZ_BLOCK) void k_act_env(Dev d, EnvDev e, double temperature, const double *u,
                                                       int32_t *out_action, double *out_probs,
                                                       double *out_root_value, float *out_obs_before,
                                                       float *out_reward, uint8_t *out_done,
                                                       int32_t *out_to_play)
{
    int t, a;
    if (!group_ids<LANES>(d, &t, &a)) return;
    const int A = d.L.A, OS = e.obs_size;
    bool on = a < A;
    double prob, qout, rv;
    int action = to_act_rule<RL, GUMBEL>(d, t, a, on, temperature, u ? u[t] : 0.0, &prob, &qout, &rv);
    if (on) out_probs[(long long)t * A + a] = prob;
    if (a != 0) return;
    out_action[t] = action;
    out_root_value[t] = rv;
    out_to_play[t] = e.to_play[t];
    float *cur = e.cur_obs + (long long)t * OS, *srch = e.search_obs + (long long)t * OS;
    uint8_t *legal = e.legal + (long long)t * A;
    if (out_obs_before) for (int i = 0; i < OS; i++) out_obs_before[(long long)t * OS + i] = cur[i];
    // the reference searches the NEXT move from the observation saved with this one (self_play.py:86-91,108)
    if (e.stale) for (int i = 0; i < OS; i++) srch[i] = cur[i];
    float reward;
    bool term;
    if (e.kind == 1) {
        signed char *b = reinterpret_cast<signed char *>(e.state) + (long long)t * 9;
        int mover = e.mover[t];
        term = tictactoe_step(b, &mover, action, &reward);
        int moves = e.moves[t] + 1;
        bool done = term || moves > e.max_moves;
        if (done) {  // BoardGame.reset (games/game.py:65-71)
            for (int i = 0; i < 9; i++) b[i] = 0;
            mover = -1; moves = 0;
        }
        e.mover[t] = mover; e.moves[t] = moves;
        e.to_play[t] = mover == -1 ? 0 : 1;
        tictactoe_observe(b, mover, cur, legal);
        if (done || !e.stale) for (int i = 0; i < OS; i++) srch[i] = cur[i];
        out_done[t] = done;
    } else {
        double *s = reinterpret_cast<double *>(e.state) + (long long)t * 4;
        term = cartpole_step(s, action);
        reward = 1.0f;
        int moves = e.moves[t] + 1;
        bool done = term || moves > e.max_moves;
        if (done) {
            for (int i = 0; i < 4; i++) s[i] = e.reset_state[(long long)t * 4 + i];
            moves = 0;
        }
        e.moves[t] = moves;
        for (int i = 0; i < 4; i++) cur[i] = (float)s[i];
        if (done || !e.stale) for (int i = 0; i < OS; i++) srch[i] = cur[i];
        out_done[t] = done;
    }
    out_reward[t] = reward;
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
__global__ void k_node_backward(Dev d, int t, int leaf, int to_play)
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
        pw.cvis[a] += 1;
        leaf_value = reward + d.discount * leaf_value;
        leaf = p;
    }
}

__global__ void k_stats(Dev d, long long *out_node_visits, int32_t *out_noise_calls)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= d.B) return;
    if (out_node_visits) out_node_visits[t] = d.visit_total[t];
    if (out_noise_calls) out_noise_calls[t] = (int32_t)d.noise_calls[t];
}

// ============================================================================ host side

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
            return fail(MZ_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));          \
    } while (0)

struct mz_ctx {
    mz_config cfg;
    Dev d;
    int lanes, rl;    // lanes per tree (group size) and action lanes (smallest power of two >= A, >= 2)
    int n_expanded;   // nodes 1..n_expanded exist (simulations done since reset)
    int selected;     // 1 between mz_select and mz_expand_backup
    int B;            // batch of the last reset
    std::vector<void *> allocs;
    double *node_tmp;   // device scratch for per-node ops [A]
    int32_t *node_act;  // device scratch
    NetDev net;         // uploaded dense-program network (mz_set_network)
    int net_scratch;    // activation floats per tree
    void *net_meta_d, *net_data_d;
    RootNetDev root;    // representation + prediction programs (mz_set_root_network)
    void *root_bufs[4];
    MfmaNet mnet;       // MFMA-layout network (mz_set_network_mfma)
    void *mnet_bufs[5];
    float *leaf_logits; // [max_batch][A] outputs of the batched leaf evaluation
    double *leaf_value, *leaf_reward;
};

static int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }
static bool shape_supported(int lanes, int rl) { return lanes == rl || (lanes == 16 && rl < 16) || (lanes == 64 && rl < 64); }

static void considered_visits_host(int m, int S, int32_t *out)
{
    if (m <= 1) { for (int i = 0; i < S; i++) out[i] = i; return; }
    int log2max = (int)std::ceil(std::log2((double)m));
    std::vector<int32_t> visits(m, 0);
    int n_considered = m, len = 0;
    while (len < S) {
        int n_extra = (int)((double)S / (double)(log2max * n_considered));
        if (n_extra < 1) n_extra = 1;
        for (int e = 0; e < n_extra && len < S; e++) {
            for (int i = 0; i < n_considered && len < S; i++) out[len++] = visits[i];
            for (int i = 0; i < n_considered; i++) visits[i]++;
        }
        n_considered = n_considered / 2 > 2 ? n_considered / 2 : 2;
    }
}

extern "C" int mz_considered_visits(int32_t m, int32_t S, int32_t *out_h)
{
    if (S < 0 || m > MZ_MAX_ACTIONS || !out_h) return fail(MZ_ERR_ARG, "mz_considered_visits: bad argument");
    considered_visits_host(m, S, out_h);
    return MZ_OK;
}

extern "C" const char *mz_last_error(void) { return g_err.c_str(); }
extern "C" int mz_version(void) { return 1; }

template <typename T>
static int dev_alloc(mz_ctx *c, T **p, size_t n)
{
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, n * sizeof(T) ? n * sizeof(T) : 16));
    c->allocs.push_back(q);
    *p = reinterpret_cast<T *>(q);
    return MZ_OK;
}

static int round_up(int x, int m) { return (x + m - 1) / m * m; }

extern "C" mz_ctx *mz_create(const mz_config *cfg)
{
    if (!cfg) { fail(MZ_ERR_ARG, "mz_create: null config"); return nullptr; }
    const int A = cfg->n_actions, S = cfg->n_simulations;
    if (A < 1 || A > MZ_MAX_ACTIONS) { fail(MZ_ERR_ARG, "mz_create: n_actions must be in 1..64"); return nullptr; }
    if (S < 1 || S > (1 << 20)) { fail(MZ_ERR_ARG, "mz_create: n_simulations out of range"); return nullptr; }
    if (cfg->players != 1 && cfg->players != 2) { fail(MZ_ERR_ARG, "mz_create: players must be 1 or 2"); return nullptr; }
    if (cfg->max_batch < 1 || cfg->hidden_size < 0) { fail(MZ_ERR_ARG, "mz_create: bad max_batch/hidden_size"); return nullptr; }
    int lanes = cfg->lanes_per_tree;
    int l0 = next_pow2(A) < 2 ? 2 : next_pow2(A);
    if (lanes == 0) {
        // Small batches: give each tree a wider group (more lanes for leaf evaluation, more waves in flight).
        lanes = l0;
        if (cfg->max_batch <= 1024 && l0 < 64) lanes = 64;
        else if (cfg->max_batch <= 32768 && l0 < 16) lanes = 16;
    }
    if (!shape_supported(lanes, l0)) {
        fail(MZ_ERR_ARG, "mz_create: lanes_per_tree must be the smallest power of two >= n_actions (min 2), or 16, or 64");
        return nullptr;
    }
    if (hipSetDevice(cfg->device) != hipSuccess) { fail(MZ_ERR_HIP, "mz_create: hipSetDevice failed"); return nullptr; }
    mz_ctx *c = new mz_ctx();
    c->cfg = *cfg;
    c->cfg.c_table_h = nullptr;
    c->lanes = lanes;
    c->n_expanded = 0; c->selected = 0; c->B = 0;
    memset(&c->net, 0, sizeof c->net); c->net_scratch = 0; c->net_meta_d = c->net_data_d = nullptr;
    memset(&c->root, 0, sizeof c->root); memset(c->root_bufs, 0, sizeof c->root_bufs);
    memset(&c->mnet, 0, sizeof c->mnet); memset(c->mnet_bufs, 0, sizeof c->mnet_bufs);
    Dev &d = c->d;
    memset(&d, 0, sizeof d);
    Layout &L = d.L;
    L.A = A; L.S = S; L.N = S + 1; L.H = cfg->hidden_size;
    L.prior_f32 = cfg->gumbel ? 1 : 0;
    L.off_cval = 32;
    L.off_crew = L.off_cval + 8 * A;
    L.off_prior = L.off_crew + 8 * A;
    L.off_cvis = L.off_prior + round_up((cfg->gumbel ? 4 : 8) * A, 8);
    L.off_cidx = L.off_cvis + 4 * A;
    L.rec = round_up(L.off_cidx + 4 * A, 64);
    L.tree_stride = (long long)L.N * L.rec;
    const size_t B = (size_t)cfg->max_batch;
    int rc = MZ_OK;
    rc |= dev_alloc(c, &d.nodes, B * (size_t)L.tree_stride);
    rc |= dev_alloc(c, &d.hidden, B * L.N * (size_t)L.H);
    rc |= dev_alloc(c, &d.legal, B);
    rc |= dev_alloc(c, &d.gumbel, B * A);
    rc |= dev_alloc(c, &d.m_considered, B);
    rc |= dev_alloc(c, &d.sel_parent, B);
    rc |= dev_alloc(c, &d.sel_action, B);
    rc |= dev_alloc(c, &d.sel_ntp, B);
    rc |= dev_alloc(c, &d.sel_depth, B);
    rc |= dev_alloc(c, &d.noise_calls, B);
    rc |= dev_alloc(c, &d.visit_total, B);
    rc |= dev_alloc(c, &d.path, B * S);
    rc |= dev_alloc(c, &d.err, (size_t)1);
    rc |= dev_alloc(c, &c->node_tmp, (size_t)MZ_MAX_ACTIONS);
    rc |= dev_alloc(c, &c->node_act, (size_t)1);
    rc |= dev_alloc(c, &c->leaf_logits, B * A);
    rc |= dev_alloc(c, &c->leaf_value, B);
    rc |= dev_alloc(c, &c->leaf_reward, B);
    int32_t *cvt = nullptr;
    double *ct = nullptr;
    rc |= dev_alloc(c, &cvt, (size_t)(A + 1) * S);
    rc |= dev_alloc(c, &ct, (size_t)S + 2);
    if (rc != MZ_OK) { mz_destroy(c); return nullptr; }
    std::vector<int32_t> tab((size_t)(A + 1) * S);
    for (int m = 0; m <= A; m++) considered_visits_host(m, S, tab.data() + (size_t)m * S);
    std::vector<double> ctab(S + 2);
    for (int n = 0; n < S + 2; n++)
        ctab[n] = cfg->c_table_h ? cfg->c_table_h[n]
                                 : cfg->c_init + std::log(((double)n + cfg->c_base + 1.0) / cfg->c_base);
    if (hipMemcpy(cvt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ct, ctab.data(), ctab.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(d.err, 0, 4) != hipSuccess || hipMemset(d.sel_depth, 0xff, B * 4) != hipSuccess) {
        fail(MZ_ERR_HIP, "mz_create: table upload failed");
        mz_destroy(c);
        return nullptr;
    }
    d.cv_table = cvt; d.c_table = ct;
    if (!cfg->gumbel && (S + 2) * (S + 1) <= MZ_SQ_LDS) {
        std::vector<double> sq((size_t)(S + 2) * (S + 1));
        for (int n = 0; n < S + 2; n++)
            for (int v = 0; v < S + 1; v++) sq[(size_t)n * (S + 1) + v] = std::sqrt((double)n / (double)(1 + v));
        double *sqd = nullptr;
        if (dev_alloc(c, &sqd, sq.size()) != MZ_OK ||
            hipMemcpy(sqd, sq.data(), sq.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
            fail(MZ_ERR_HIP, "mz_create: sqrt table upload failed");
            mz_destroy(c);
            return nullptr;
        }
        d.sqrt_table = sqd;
    }
    d.players = cfg->players; d.noise_mode = cfg->noise_mode;
    d.RL = l0;
    c->rl = l0;
    d.discount = cfg->discount; d.c_visit = cfg->c_visit; d.c_scale = cfg->c_scale;
    d.seed = cfg->seed;
    return c;
}

extern "C" void mz_destroy(mz_ctx *c)
{
    if (!c) return;
    for (void *p : c->allocs) (void)hipFree(p);
    if (c->net_meta_d) (void)hipFree(c->net_meta_d);
    if (c->net_data_d) (void)hipFree(c->net_data_d);
    for (void *p : c->root_bufs) if (p) (void)hipFree(p);
    for (void *p : c->mnet_bufs) if (p) (void)hipFree(p);
    delete c;
}

// Calls f(lanes tag, action-lanes tag, gumbel tag) for the (lanes per tree, action lanes, rule) of a context.
// Supported shapes: a group is either exactly the action lanes, or 16 or 64 lanes.
template <int N> using ic = std::integral_constant<int, N>;
template <typename F>
static bool dispatch_shape(int lanes, int rl, bool gumbel, F &&f)
{
#define MZ_SHAPE(L, R)                                                     \
    if (lanes == L && rl == R) {                                           \
        if (gumbel) f(ic<L>(), ic<R>(), std::true_type());                 \
        else f(ic<L>(), ic<R>(), std::false_type());                       \
        return true;                                                       \
    }
    MZ_SHAPE(2, 2) MZ_SHAPE(16, 2) MZ_SHAPE(64, 2)
    MZ_SHAPE(4, 4) MZ_SHAPE(16, 4) MZ_SHAPE(64, 4)
    MZ_SHAPE(8, 8) MZ_SHAPE(16, 8) MZ_SHAPE(64, 8)
    MZ_SHAPE(16, 16) MZ_SHAPE(64, 16)
    MZ_SHAPE(32, 32) MZ_SHAPE(64, 32)
    MZ_SHAPE(64, 64)
#undef MZ_SHAPE
    return false;
}

static unsigned grid_for(long long groups, int lanes) { return (unsigned)((groups * lanes + MZ_BLOCK - 1) / MZ_BLOCK); }

// Heavy kernels run with the context's group size; light once-per-search kernels with group = action lanes.
#define MZ_LAUNCH(c, KERNEL, light, groups, lds_bytes, stream, ...)                                          \
    dispatch_shape((light) ? (c)->rl : (c)->lanes, (c)->rl, (c)->cfg.gumbel != 0, [&](auto L_, auto R_, auto G_) { \
        hipLaunchKernelGGL((KERNEL<decltype(L_)::value, decltype(R_)::value, decltype(G_)::value>),           \
                           dim3(grid_for((groups), decltype(L_)::value)), dim3(MZ_BLOCK), (lds_bytes),        \
                           (hipStream_t)(stream), __VA_ARGS__);                                               \
    })

static int check_ctx(mz_ctx *c, const char *fn)
{
    if (!c) return fail(MZ_ERR_ARG, std::string(fn) + ": null context");
    if (hipSetDevice(c->cfg.device) != hipSuccess) return fail(MZ_ERR_HIP, std::string(fn) + ": hipSetDevice failed");
    (void)hipGetLastError();  // a stale error left by other users of the runtime is not ours to report
    return MZ_OK;
}

static int launch_status(const char *fn)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MZ_ERR_HIP, std::string(fn) + ": launch failed: " + hipGetErrorString(e));
    return MZ_OK;
}

extern "C" int mz_root_prepare(mz_ctx *c, int32_t B, const float *logits, const double *dirichlet,
                               double frac, const uint8_t *legal, int32_t shift_and_mask, void *out_prior,
                               void *stream)
{
    int rc = check_ctx(c, "mz_root_prepare");
    if (rc) return rc;
    if (B < 1 || !logits || !out_prior) return fail(MZ_ERR_ARG, "mz_root_prepare: null/empty argument");
    if (!legal && (shift_and_mask || c->cfg.gumbel)) return fail(MZ_ERR_ARG, "mz_root_prepare: legal mask required");
    if (!c->cfg.gumbel && !dirichlet) return fail(MZ_ERR_ARG, "mz_root_prepare: PUCT needs dirichlet noise");
    MZ_LAUNCH(c, k_root_prepare, true, B, 0, stream, (int)B, c->cfg.n_actions, logits, dirichlet, frac, legal,
                (int)shift_and_mask, out_prior);
    return launch_status("mz_root_prepare");
}

extern "C" int mz_tree_reset(mz_ctx *c, int32_t B, const void *root_prior, const double *root_value,
                             const float *root_hidden, const int32_t *root_to_play, const uint8_t *legal,
                             const double *gumbel, uint64_t tree_id0, void *stream)
{
    int rc = check_ctx(c, "mz_tree_reset");
    if (rc) return rc;
    if (B < 1 || B > c->cfg.max_batch) return fail(MZ_ERR_ARG, "mz_tree_reset: B outside 1..max_batch");
    if (!root_prior || !root_value || !root_to_play || !legal) return fail(MZ_ERR_ARG, "mz_tree_reset: null argument");
    if (c->cfg.gumbel && !gumbel) return fail(MZ_ERR_ARG, "mz_tree_reset: Gumbel search needs gumbel noise");
    c->B = B; c->d.B = B; c->d.tree_id0 = tree_id0;
    c->n_expanded = 0; c->selected = 0;
    MZ_LAUNCH(c, k_reset, true, B, 0, stream, c->d, root_prior, root_value, root_hidden, root_to_play, legal, gumbel,
                (int)c->cfg.max_considered_actions);
    return launch_status("mz_tree_reset");
}

static int check_noise(mz_ctx *c, const double *noise, int32_t max_calls, const char *fn)
{
    if (!c->cfg.gumbel && c->cfg.noise_mode == MZ_NOISE_BUFFER && (!noise || max_calls < 1))
        return fail(MZ_ERR_ARG, std::string(fn) + ": noise_mode is MZ_NOISE_BUFFER but no noise buffer given");
    return MZ_OK;
}

extern "C" int mz_select(mz_ctx *c, const double *noise, int32_t noise_max_calls, float *out_parent_hidden,
                         int32_t *out_parent, int32_t *out_action, void *stream)
{
    int rc = check_ctx(c, "mz_select");
    if (rc) return rc;
    if (c->B < 1) return fail(MZ_ERR_STATE, "mz_select: no tree (call mz_tree_reset first)");
    if (c->n_expanded >= c->cfg.n_simulations) return fail(MZ_ERR_STATE, "mz_select: all simulations already done");
    if ((rc = check_noise(c, noise, noise_max_calls, "mz_select"))) return rc;
    if (c->cfg.noise_mode != MZ_NOISE_BUFFER) noise = nullptr;
    MZ_LAUNCH(c, k_select, false, c->B, 0, stream, c->d, noise, (int)noise_max_calls, out_parent_hidden, out_parent,
                out_action);
    c->selected = 1;
    return launch_status("mz_select");
}

extern "C" int mz_select_outputs(mz_ctx *c, const int32_t **parent, const int32_t **action,
                                 const int32_t **ntp, const int32_t **depth)
{
    if (!c) return fail(MZ_ERR_ARG, "mz_select_outputs: null context");
    if (parent) *parent = c->d.sel_parent;
    if (action) *action = c->d.sel_action;
    if (ntp) *ntp = c->d.sel_ntp;
    if (depth) *depth = c->d.sel_depth;
    return MZ_OK;
}

extern "C" int mz_expand_backup(mz_ctx *c, const float *logits, const double *value, const double *reward,
                                const float *next_hidden, void *stream)
{
    int rc = check_ctx(c, "mz_expand_backup");
    if (rc) return rc;
    if (!c->selected) return fail(MZ_ERR_STATE, "mz_expand_backup: no pending selection (call mz_select first)");
    if (!logits || !value || !reward) return fail(MZ_ERR_ARG, "mz_expand_backup: null argument");
    int leaf = c->n_expanded + 1;
    MZ_LAUNCH(c, k_expand_backup, false, c->B, 0, stream, c->d, leaf, logits, value, reward, next_hidden);
    c->n_expanded = leaf;
    c->selected = 0;
    return launch_status("mz_expand_backup");
}

extern "C" int mz_search_run(mz_ctx *c, int32_t net_kind, const void *net_params_h, const double *noise,
                             int32_t noise_max_calls, void *stream)
{
    int rc = check_ctx(c, "mz_search_run");
    if (rc) return rc;
    if (c->B < 1) return fail(MZ_ERR_STATE, "mz_search_run: no tree (call mz_tree_reset first)");
    if (c->selected) return fail(MZ_ERR_STATE, "mz_search_run: a selection is pending");
    if (net_kind != MZ_NET_HASH && net_kind != MZ_NET_DENSE && net_kind != MZ_NET_MFMA)
        return fail(MZ_ERR_ARG, "mz_search_run: unsupported net_kind");
    if ((rc = check_noise(c, noise, noise_max_calls, "mz_search_run"))) return rc;
    if (c->cfg.noise_mode != MZ_NOISE_BUFFER) noise = nullptr;
    int n_sims = c->cfg.n_simulations - c->n_expanded;
    if (n_sims <= 0) return fail(MZ_ERR_STATE, "mz_search_run: all simulations already done");
    if (net_kind == MZ_NET_MFMA) {
        if (!c->mnet.steps) return fail(MZ_ERR_STATE, "mz_search_run: no network uploaded (call mz_set_network_mfma first)");
        if (c->rl > 16) return fail(MZ_ERR_ARG, "mz_search_run: the matrix-core search serves at most 16 actions; use the step API");
        constexpr int tm = MZ_NET_TM;  // 16-tree tiles (two blocks per CU) measured no faster on the 3x3-board ResNet
        const size_t lds = (size_t)c->mnet.stride * tm * 4;
        if (lds + (size_t)tm * 16 * 8 > 160 * 1024)
            return fail(MZ_ERR_ARG, "mz_search_run: activations of a tile do not fit in LDS beside the backup staging; use the step API");
        hipError_t attr_err = hipSuccess;
        auto launch = [&](auto kern) {
            attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
            hipLaunchKernelGGL(kern, dim3((c->B + tm - 1) / tm), dim3(tm * 16), lds, (hipStream_t)stream,
                               c->d, c->mnet, c->n_expanded + 1, n_sims, noise, (int)noise_max_calls);
        };
        const bool g = c->cfg.gumbel != 0;
        switch (c->rl) {
        case 2: if (g) launch(k_search_mfma<tm, 2, true>); else launch(k_search_mfma<tm, 2, false>); break;
        case 4: if (g) launch(k_search_mfma<tm, 4, true>); else launch(k_search_mfma<tm, 4, false>); break;
        case 8: if (g) launch(k_search_mfma<tm, 8, true>); else launch(k_search_mfma<tm, 8, false>); break;
        default: if (g) launch(k_search_mfma<tm, 16, true>); else launch(k_search_mfma<tm, 16, false>); break;
        }
        if (attr_err != hipSuccess) return fail(MZ_ERR_HIP, "mz_search_run: could not raise the dynamic LDS limit");
        c->n_expanded += n_sims;
        return launch_status("mz_search_run");
    }
    if (net_kind == MZ_NET_DENSE) {
        if (!c->net.meta) return fail(MZ_ERR_STATE, "mz_search_run: no network uploaded (call mz_set_network first)");
        size_t lds = ((size_t)c->net.n_data + c->net.n_meta + (size_t)(MZ_BLOCK / c->lanes) * c->net_scratch) * 4;
        if (lds > 160 * 1024)
            return fail(MZ_ERR_ARG, "mz_search_run: network does not fit in LDS for the fused search; use the step API");
        // Small batches are latency-bound: whole-wave groups keep their trees in LDS when one block per CU
        // covers the batch and everything fits beside the kernel's static tables.
        const Layout &L = c->d.L;
        const int tree_lds = ((L.N * L.rec + 15) & ~15) + ((L.N * L.H * 4 + 15) & ~15) + ((L.S * 4 + 15) & ~15) +
                             ((L.N * 4 + 15) & ~15);
        const size_t lds_tree = ((lds + 15) & ~(size_t)15) + (size_t)(MZ_BLOCK / c->lanes) * tree_lds;
        const size_t lds_static = (size_t)MZ_BLOCK * 8 + (c->cfg.gumbel ? 32 : (size_t)(MZ_CT_LDS + MZ_SQ_LDS) * 8);
        bool tree_in_lds = c->lanes == 64 && grid_for(c->B, 64) <= 256 && lds_tree + lds_static <= 160 * 1024;
        if (const char *e = getenv("MZ_TREE_LDS")) tree_in_lds = tree_in_lds && atoi(e) != 0;
        hipError_t attr_err = hipSuccess;
        dispatch_shape(c->lanes, c->rl, c->cfg.gumbel != 0, [&](auto L_, auto R_, auto G_) {
            constexpr int LN = decltype(L_)::value, RN = decltype(R_)::value;
            constexpr bool GM = decltype(G_)::value;
            auto launch = [&](auto kern, size_t bytes, int tree_bytes) {
                attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
                hipLaunchKernelGGL(kern, dim3(grid_for(c->B, LN)), dim3(MZ_BLOCK), bytes, (hipStream_t)stream, c->d,
                                   c->net, c->n_expanded + 1, n_sims, noise, (int)noise_max_calls, c->net_scratch,
                                   tree_bytes);
            };
            if constexpr (LN == 64) {
                if (tree_in_lds) { launch(k_search_net<LN, RN, GM, true>, lds_tree, tree_lds); return; }
            }
            launch(k_search_net<LN, RN, GM, false>, lds, 0);
        });
        if (attr_err != hipSuccess) return fail(MZ_ERR_HIP, "mz_search_run: could not raise the dynamic LDS limit");
        c->n_expanded += n_sims;
        return launch_status("mz_search_run");
    }
    if (!net_params_h) return fail(MZ_ERR_ARG, "mz_search_run: null network parameters");
    if (c->cfg.hidden_size < 2) return fail(MZ_ERR_ARG, "mz_search_run: hash network needs hidden_size >= 2");
    const int32_t *p = reinterpret_cast<const int32_t *>(net_params_h);
    HashNet hn{p[0], p[1], p[2]};
    MZ_LAUNCH(c, k_search_hash, false, c->B, 0, stream, c->d, hn, c->n_expanded + 1, n_sims, noise, (int)noise_max_calls);
    c->n_expanded += n_sims;
    return launch_status("mz_search_run");
}

extern "C" int mz_to_act(mz_ctx *c, double temperature, const double *u, int32_t *out_action,
                         double *out_probs, double *out_root_value, double *out_q, void *stream)
{
    int rc = check_ctx(c, "mz_to_act");
    if (rc) return rc;
    if (c->B < 1) return fail(MZ_ERR_STATE, "mz_to_act: no tree");
    if (!out_action || !out_probs || !out_root_value) return fail(MZ_ERR_ARG, "mz_to_act: null output");
    MZ_LAUNCH(c, k_to_act, true, c->B, 0, stream, c->d, temperature, u, out_action, out_probs, out_root_value, out_q);
    return launch_status("mz_to_act");
}

extern "C" int mz_make_targets(const mz_trajectories *tr, int32_t n_samples, const int32_t *game_idx, const int32_t *t,
                               const int32_t *absorbed, int32_t td_steps, int32_t unroll_steps, double gamma,
                               const double *gamma_pow, int32_t n_stack, int32_t stack_action, float *out_obs,
                               float *out_actions, float *out_value, float *out_reward, float *out_policy,
                               void *stream)
{
    (void)hipGetLastError();
    if (!tr || n_samples < 1 || !game_idx || !t || !absorbed || !out_obs || !out_actions || !out_value || !out_reward ||
        !out_policy)
        return fail(MZ_ERR_ARG, "mz_make_targets: null/empty argument");
    if (!tr->game_offset || !tr->observations || !tr->actions || !tr->rewards || !tr->to_plays || !tr->root_values ||
        !tr->probs || tr->n_games < 1 || tr->obs_size < 1 || tr->n_actions < 1)
        return fail(MZ_ERR_ARG, "mz_make_targets: incomplete trajectory store");
    if (td_steps < 0 || unroll_steps < 0 || n_stack < 1) return fail(MZ_ERR_ARG, "mz_make_targets: bad step counts");
    if (gamma != 1.0 && !gamma_pow) return fail(MZ_ERR_ARG, "mz_make_targets: gamma_pow table required when gamma != 1");
    TrajDev d;
    d.n_games = tr->n_games; d.obs_size = tr->obs_size; d.A = tr->n_actions;
    d.game_offset = reinterpret_cast<const long long *>(tr->game_offset);
    d.observations = tr->observations; d.actions = tr->actions; d.rewards = tr->rewards; d.to_plays = tr->to_plays;
    d.root_values = tr->root_values; d.probs = tr->probs;
    d.re_root_values = tr->re_root_values; d.re_probs = tr->re_probs; d.has_re = tr->has_reanalysed;
    long long n = (long long)n_samples * (unroll_steps + 1);
    hipLaunchKernelGGL(k_make_targets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d,
                       (int)n_samples, game_idx, t, absorbed, (int)td_steps, (int)unroll_steps, gamma, gamma_pow,
                       (int)n_stack, (int)stack_action, out_obs, out_actions, out_value, out_reward, out_policy);
    return launch_status("mz_make_targets");
}

extern "C" int mz_stats(mz_ctx *c, int64_t *out_node_visits, int32_t *out_noise_calls, void *stream)
{
    int rc = check_ctx(c, "mz_stats");
    if (rc) return rc;
    if (c->B < 1) return fail(MZ_ERR_STATE, "mz_stats: no tree");
    hipLaunchKernelGGL(k_stats, dim3((c->B + 255) / 256), dim3(256), 0, (hipStream_t)stream, c->d,
                       reinterpret_cast<long long *>(out_node_visits), out_noise_calls);
    return launch_status("mz_stats");
}

static int device_error(mz_ctx *c, hipStream_t st, const char *fn)
{
    int32_t e = 0;
    HIPCHK(hipMemcpyAsync(&e, c->d.err, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (e) {
        (void)hipMemsetAsync(c->d.err, 0, 4, st);
        return fail(MZ_ERR_DEVICE, std::string(fn) + ": a kernel reported an inconsistent selection state");
    }
    return MZ_OK;
}

extern "C" int mz_tree_export(mz_ctx *c, int32_t b, mz_tree_host *o, void *stream)
{
    int rc = check_ctx(c, "mz_tree_export");
    if (rc) return rc;
    if (!o || b < 0 || b >= c->B) return fail(MZ_ERR_ARG, "mz_tree_export: tree index out of range");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = device_error(c, st, "mz_tree_export"))) return rc;
    const Layout &L = c->d.L;
    const int A = L.A, N = L.N, H = L.H;
    std::vector<char> buf((size_t)L.tree_stride);
    HIPCHK(hipMemcpyAsync(buf.data(), c->d.nodes + (long long)b * L.tree_stride, buf.size(), hipMemcpyDeviceToHost, st));
    if (o->hidden_states && H > 0)
        HIPCHK(hipMemcpyAsync(o->hidden_states, c->d.hidden + (long long)b * N * H, (size_t)N * H * 4,
                              hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int n = 0; n < N; n++) {
        const char *r = buf.data() + (size_t)n * L.rec;
        bool live = n <= c->n_expanded;
        NodeHdr h;
        memcpy(&h, r, sizeof h);
        o->node_visits[n] = live ? h.visits : 0;
        o->node_values[n] = live ? h.value : 0.0;
        o->node_raw_values[n] = live ? h.raw : 0.0;
        o->parents[n] = live ? h.parent : -1;
        o->action_from_parent[n] = live ? h.afp : -1;
        o->to_play[n] = live ? h.to_play : 0;
        for (int a = 0; a < A; a++) {
            size_t k = (size_t)n * A + a;
            if (live) {
                memcpy(&o->children_values[k], r + L.off_cval + 8 * a, 8);
                memcpy(&o->children_rewards[k], r + L.off_crew + 8 * a, 8);
                memcpy(&o->children_visits[k], r + L.off_cvis + 4 * a, 4);
                memcpy(&o->children_index[k], r + L.off_cidx + 4 * a, 4);
                if (L.prior_f32) memcpy((float *)o->children_prior_logits + k, r + L.off_prior + 4 * a, 4);
                else memcpy((double *)o->children_prior_logits + k, r + L.off_prior + 8 * a, 8);
            } else {
                o->children_values[k] = 0; o->children_rewards[k] = 0;
                o->children_visits[k] = 0; o->children_index[k] = -1;
                if (L.prior_f32) ((float *)o->children_prior_logits)[k] = 0;
                else ((double *)o->children_prior_logits)[k] = 0;
            }
        }
        if (!live && o->hidden_states && H > 0) memset(o->hidden_states + (size_t)n * H, 0, (size_t)H * 4);
    }
    return MZ_OK;
}

extern "C" int mz_tree_import(mz_ctx *c, int32_t b, const mz_tree_host *in, const double *root_legal_h,
                              const double *gumbel_h, void *stream)
{
    int rc = check_ctx(c, "mz_tree_import");
    if (rc) return rc;
    if (!in || !root_legal_h || b < 0 || b >= c->cfg.max_batch) return fail(MZ_ERR_ARG, "mz_tree_import: bad argument");
    if (c->cfg.gumbel && !gumbel_h) return fail(MZ_ERR_ARG, "mz_tree_import: Gumbel tree needs gumbel noise");
    hipStream_t st = (hipStream_t)stream;
    const Layout &L = c->d.L;
    const int A = L.A, N = L.N, H = L.H;
    std::vector<char> buf((size_t)L.tree_stride, 0);
    int live = 0;
    for (int n = 0; n < N; n++) {
        char *r = buf.data() + (size_t)n * L.rec;
        NodeHdr h;
        h.visits = in->node_visits[n]; h.parent = in->parents[n]; h.afp = in->action_from_parent[n];
        h.to_play = in->to_play[n]; h.value = in->node_values[n]; h.raw = in->node_raw_values[n];
        if (h.visits > 0) live = n;
        memcpy(r, &h, sizeof h);
        for (int a = 0; a < A; a++) {
            size_t k = (size_t)n * A + a;
            memcpy(r + L.off_cval + 8 * a, &in->children_values[k], 8);
            memcpy(r + L.off_crew + 8 * a, &in->children_rewards[k], 8);
            memcpy(r + L.off_cvis + 4 * a, &in->children_visits[k], 4);
            memcpy(r + L.off_cidx + 4 * a, &in->children_index[k], 4);
            if (L.prior_f32) memcpy(r + L.off_prior + 4 * a, (const float *)in->children_prior_logits + k, 4);
            else memcpy(r + L.off_prior + 8 * a, (const double *)in->children_prior_logits + k, 8);
        }
    }
    unsigned long long mask = 0;
    int n_legal = 0;
    for (int a = 0; a < A; a++) if (root_legal_h[a] != 0) { mask |= 1ull << a; n_legal++; }
    int32_t m = c->cfg.max_considered_actions < n_legal ? c->cfg.max_considered_actions : n_legal;
    HIPCHK(hipMemcpyAsync(c->d.nodes + (long long)b * L.tree_stride, buf.data(), buf.size(), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->d.legal + b, &mask, 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->d.m_considered + b, &m, 4, hipMemcpyHostToDevice, st));
    if (gumbel_h) HIPCHK(hipMemcpyAsync(c->d.gumbel + (long long)b * A, gumbel_h, (size_t)A * 8, hipMemcpyHostToDevice, st));
    if (in->hidden_states && H > 0)
        HIPCHK(hipMemcpyAsync(c->d.hidden + (long long)b * N * H, in->hidden_states, (size_t)N * H * 4,
                              hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    if (b >= c->B) { c->B = b + 1; c->d.B = c->B; }
    if (live > c->n_expanded) c->n_expanded = live;
    return MZ_OK;
}

static int node_op(mz_ctx *c, int32_t b, int32_t node, int depth, int which, const double *uniform_h,
                   double *out_vec_h, int32_t *out_action_h, void *stream, const char *fn)
{
    int rc = check_ctx(c, fn);
    if (rc) return rc;
    if (b < 0 || b >= c->B || node < 0 || node >= c->d.L.N) return fail(MZ_ERR_ARG, std::string(fn) + ": tree/node out of range");
    hipStream_t st = (hipStream_t)stream;
    const int A = c->d.L.A;
    double *u_d = nullptr;
    if (uniform_h) {
        u_d = c->node_tmp;
        HIPCHK(hipMemcpyAsync(u_d, uniform_h, (size_t)A * 8, hipMemcpyHostToDevice, st));
    }
    double *vec_d = c->node_tmp;
    dispatch_shape(c->rl, c->rl, c->cfg.gumbel != 0, [&](auto L_, auto R_, auto G_) {
        hipLaunchKernelGGL((k_node_op<decltype(L_)::value, decltype(R_)::value, decltype(G_)::value>), dim3(1), dim3(64),
                           0, st, c->d, (int)b, (int)node, depth, which, u_d, vec_d, c->node_act);
    });
    if ((rc = launch_status(fn))) return rc;
    if (out_vec_h) HIPCHK(hipMemcpyAsync(out_vec_h, vec_d, (size_t)A * 8, hipMemcpyDeviceToHost, st));
    if (out_action_h) HIPCHK(hipMemcpyAsync(out_action_h, c->node_act, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return MZ_OK;
}

extern "C" int mz_node_qtransform(mz_ctx *c, int32_t b, int32_t node, int32_t which, double *out_h, void *stream)
{
    if (which < 0 || which > 2 || !out_h) return fail(MZ_ERR_ARG, "mz_node_qtransform: bad argument");
    if (c && which > 0 && !c->cfg.gumbel)
        return fail(MZ_ERR_ARG, "mz_node_qtransform: completed-Q transforms need a Gumbel context (float32 priors)");
    return node_op(c, b, node, 1, which, nullptr, out_h, nullptr, stream, "mz_node_qtransform");
}

extern "C" int mz_node_select(mz_ctx *c, int32_t b, int32_t node, int32_t depth, const double *uniform_h,
                              int32_t *out_action_h, void *stream)
{
    if (!out_action_h) return fail(MZ_ERR_ARG, "mz_node_select: null output");
    return node_op(c, b, node, depth, 3, uniform_h, nullptr, out_action_h, stream, "mz_node_select");
}

extern "C" int mz_node_backward(mz_ctx *c, int32_t b, int32_t leaf, int32_t to_play, void *stream)
{
    int rc = check_ctx(c, "mz_node_backward");
    if (rc) return rc;
    if (b < 0 || b >= c->B || leaf < 0 || leaf >= c->d.L.N) return fail(MZ_ERR_ARG, "mz_node_backward: tree/node out of range");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_node_backward, dim3(1), dim3(64), 0, st, c->d, (int)b, (int)leaf, (int)to_play);
    if ((rc = launch_status("mz_node_backward"))) return rc;
    return device_error(c, st, "mz_node_backward");
}

extern "C" int mz_set_network(mz_ctx *c, const int32_t *meta_h, int32_t n_meta, const float *data_h,
                              int32_t n_data, const int32_t *io_h)
{
    int rc = check_ctx(c, "mz_set_network");
    if (rc) return rc;
    if (!meta_h || !data_h || !io_h || n_meta < 8 || n_data < 1) return fail(MZ_ERR_ARG, "mz_set_network: null/empty program");
    if (meta_h[3] != n_data || meta_h[6] + 4 * meta_h[2] != n_meta) return fail(MZ_ERR_ARG, "mz_set_network: inconsistent program header");
    if (io_h[6] != c->cfg.hidden_size || io_h[7] != c->cfg.n_actions)
        return fail(MZ_ERR_ARG, "mz_set_network: network hidden size / action count differ from the context's");
    if (io_h[10] < 1 || io_h[6] % io_h[10]) return fail(MZ_ERR_ARG, "mz_set_network: hidden size not divisible by normalisation groups");
    // every row must read and write inside the scratch
    const int32_t *lv = meta_h + meta_h[5], *rows = meta_h + meta_h[6];
    for (int l = 0; l < meta_h[0]; l++)
        for (int r = 0; r < lv[4 * l]; r++) {
            const int32_t *d = rows + 4 * (lv[4 * l + 3] + r);
            if (d[0] < 0 || d[0] + lv[4 * l + 1] > meta_h[1] || d[1] < 0 || d[1] >= meta_h[1] || d[3] >= meta_h[1])
                return fail(MZ_ERR_ARG, "mz_set_network: a row addresses outside the activation scratch");
        }
    if (c->net_meta_d) { (void)hipFree(c->net_meta_d); c->net_meta_d = nullptr; }
    if (c->net_data_d) { (void)hipFree(c->net_data_d); c->net_data_d = nullptr; }
    HIPCHK(hipMalloc(&c->net_meta_d, (size_t)n_meta * 4));
    HIPCHK(hipMalloc(&c->net_data_d, (size_t)n_data * 4));
    HIPCHK(hipMemcpy(c->net_meta_d, meta_h, (size_t)n_meta * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->net_data_d, data_h, (size_t)n_data * 4, hipMemcpyHostToDevice));
    c->net.meta = reinterpret_cast<const int32_t *>(c->net_meta_d);
    c->net.data = reinterpret_cast<const float *>(c->net_data_d);
    c->net.n_meta = n_meta; c->net.n_data = n_data;
    memcpy(c->net.io, io_h, sizeof c->net.io);
    c->net_scratch = meta_h[1];
    return MZ_OK;
}

static int check_program(const int32_t *meta_h, int32_t n_meta, const float *data_h, int32_t n_data, const char *fn)
{
    if (!meta_h || !data_h || n_meta < 8 || n_data < 1) return fail(MZ_ERR_ARG, std::string(fn) + ": null/empty program");
    if (meta_h[3] != n_data || meta_h[6] + 4 * meta_h[2] != n_meta) return fail(MZ_ERR_ARG, std::string(fn) + ": inconsistent program header");
    const int32_t *lv = meta_h + meta_h[5], *rows = meta_h + meta_h[6];
    for (int l = 0; l < meta_h[0]; l++)
        for (int r = 0; r < lv[4 * l]; r++) {
            const int32_t *d = rows + 4 * (lv[4 * l + 3] + r);
            if (d[0] < 0 || d[0] + lv[4 * l + 1] > meta_h[1] || d[1] < 0 || d[1] >= meta_h[1] || d[3] >= meta_h[1])
                return fail(MZ_ERR_ARG, std::string(fn) + ": a row addresses outside the activation scratch");
        }
    return MZ_OK;
}

extern "C" int mz_set_root_network(mz_ctx *c, const int32_t *rep_meta_h, int32_t rep_n_meta, const float *rep_data_h,
                                   int32_t rep_n_data, const int32_t *pred_meta_h, int32_t pred_n_meta,
                                   const float *pred_data_h, int32_t pred_n_data, const int32_t *io_h)
{
    int rc = check_ctx(c, "mz_set_root_network");
    if (rc) return rc;
    if (!io_h) return fail(MZ_ERR_ARG, "mz_set_root_network: null io");
    if ((rc = check_program(rep_meta_h, rep_n_meta, rep_data_h, rep_n_data, "mz_set_root_network"))) return rc;
    if ((rc = check_program(pred_meta_h, pred_n_meta, pred_data_h, pred_n_data, "mz_set_root_network"))) return rc;
    if (io_h[7] < 1 || c->cfg.hidden_size % io_h[7]) return fail(MZ_ERR_ARG, "mz_set_root_network: bad normalisation groups");
    for (void *&p : c->root_bufs) if (p) { (void)hipFree(p); p = nullptr; }
    const void *src[4] = {rep_meta_h, rep_data_h, pred_meta_h, pred_data_h};
    size_t n[4] = {(size_t)rep_n_meta, (size_t)rep_n_data, (size_t)pred_n_meta, (size_t)pred_n_data};
    for (int i = 0; i < 4; i++) {
        HIPCHK(hipMalloc(&c->root_bufs[i], n[i] * 4));
        HIPCHK(hipMemcpy(c->root_bufs[i], src[i], n[i] * 4, hipMemcpyHostToDevice));
    }
    RootNetDev &r = c->root;
    r.rep_meta = (const int32_t *)c->root_bufs[0]; r.rep_data = (const float *)c->root_bufs[1];
    r.pred_meta = (const int32_t *)c->root_bufs[2]; r.pred_data = (const float *)c->root_bufs[3];
    r.rep_n_meta = rep_n_meta; r.rep_n_data = rep_n_data; r.pred_n_meta = pred_n_meta; r.pred_n_data = pred_n_data;
    r.rep_scratch = rep_meta_h[1]; r.pred_scratch = pred_meta_h[1];
    r.obs_off = io_h[0]; r.raw_off = io_h[1]; r.hid_off = io_h[2]; r.pol_off = io_h[3]; r.val_off = io_h[4];
    r.obs_size = io_h[5]; r.support = io_h[6]; r.ngroups = io_h[7];
    return MZ_OK;
}

extern "C" int mz_search_begin_net(mz_ctx *c, int32_t B, const float *obs, const uint8_t *legal,
                                   const int32_t *to_play, const double *dirichlet, double frac,
                                   const double *gumbel, uint64_t tree_id0, float *out_logits,
                                   double *out_value, void *stream)
{
    int rc = check_ctx(c, "mz_search_begin_net");
    if (rc) return rc;
    if (!c->root.rep_meta) return fail(MZ_ERR_STATE, "mz_search_begin_net: no root network (call mz_set_root_network first)");
    if (B < 1 || B > c->cfg.max_batch) return fail(MZ_ERR_ARG, "mz_search_begin_net: B outside 1..max_batch");
    if (!obs || !legal || !to_play) return fail(MZ_ERR_ARG, "mz_search_begin_net: null argument");
    if (c->cfg.gumbel ? !gumbel : !dirichlet) return fail(MZ_ERR_ARG, "mz_search_begin_net: root noise missing");
    const RootNetDev &r = c->root;
    size_t lds = ((size_t)r.rep_n_data + r.rep_n_meta + r.pred_n_data + r.pred_n_meta +
                  (size_t)(MZ_BLOCK / c->lanes) * (r.rep_scratch + r.pred_scratch)) * 4;
    if (lds > 160 * 1024) return fail(MZ_ERR_ARG, "mz_search_begin_net: root network does not fit in LDS");
    c->B = B; c->d.B = B; c->d.tree_id0 = tree_id0;
    c->n_expanded = 0; c->selected = 0;
    hipError_t attr_err = hipSuccess;
    dispatch_shape(c->lanes, c->rl, c->cfg.gumbel != 0, [&](auto L_, auto R_, auto G_) {
        auto kern = k_begin_net<decltype(L_)::value, decltype(R_)::value, decltype(G_)::value>;
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid_for(B, decltype(L_)::value)), dim3(MZ_BLOCK), lds, (hipStream_t)stream,
                           c->d, r, obs, legal, to_play, dirichlet, frac, gumbel,
                           (int)c->cfg.max_considered_actions, out_logits, out_value);
    });
    if (attr_err != hipSuccess) return fail(MZ_ERR_HIP, "mz_search_begin_net: could not raise the dynamic LDS limit");
    return launch_status("mz_search_begin_net");
}

#ifdef MZ_STAMPS
// Diagnostic build only: copies the first n_bytes of the path buffer (where k_search_net parks its phase stamps).
extern "C" int mz_debug_read_path(mz_ctx *c, void *out_h, size_t n_bytes)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_h, c->d.path, n_bytes, hipMemcpyDeviceToHost));
    return MZ_OK;
}
#endif

extern "C" int mz_set_network_mfma(mz_ctx *c, const int32_t *steps_h, int32_t n_steps, const int32_t *tiles_h,
                                   int32_t n_tiles, const int32_t *level_start_h, int32_t n_levels,
                                   const float *wfrag_h, int64_t n_w, const float *bias_h, int32_t n_b,
                                   int32_t stride, const int32_t *io_h)
{
    int rc = check_ctx(c, "mz_set_network_mfma");
    if (rc) return rc;
    if (!steps_h || !tiles_h || !level_start_h || !wfrag_h || !bias_h || !io_h || n_steps < 1 || n_tiles < 1 || n_levels < 1)
        return fail(MZ_ERR_ARG, "mz_set_network_mfma: null/empty program");
    if (io_h[6] != c->cfg.hidden_size || io_h[7] != c->cfg.n_actions)
        return fail(MZ_ERR_ARG, "mz_set_network_mfma: network hidden size / action count differ from the context's");
    if (io_h[10] < 1 || io_h[6] % io_h[10]) return fail(MZ_ERR_ARG, "mz_set_network_mfma: hidden size not divisible by normalisation groups");
    if (stride % 64 != 4) return fail(MZ_ERR_ARG, "mz_set_network_mfma: stride must be 4 mod 64");
    if ((size_t)stride * MZ_NET_TM * 4 > 160 * 1024) return fail(MZ_ERR_ARG, "mz_set_network_mfma: activations of a 32-tree tile exceed LDS");
    if (level_start_h[n_levels] != n_tiles) return fail(MZ_ERR_ARG, "mz_set_network_mfma: inconsistent tile table");
    for (int s = 0; s < n_steps; s++) {
        const int32_t *st = steps_h + 12 * s;
        int k16 = st[2], nct = st[9];
        if (st[3] < 0 || st[3] + k16 * 16 > stride || st[4] < 0 || st[4] + nct * 16 > stride || st[5] + nct * 16 > stride ||
            st[7] < 0 || (int64_t)st[7] + (int64_t)nct * k16 * 256 > n_w || st[8] < 0 || st[8] + nct * 16 > n_b)
            return fail(MZ_ERR_ARG, "mz_set_network_mfma: a step addresses outside its buffers");
    }
    for (int i = 0; i < n_tiles; i++)
        if (tiles_h[2 * i] < 0 || tiles_h[2 * i] >= n_steps || tiles_h[2 * i + 1] < 0 ||
            tiles_h[2 * i + 1] >= steps_h[12 * tiles_h[2 * i] + 9])
            return fail(MZ_ERR_ARG, "mz_set_network_mfma: bad tile entry");
    for (void *&p : c->mnet_bufs) if (p) { (void)hipFree(p); p = nullptr; }
    const void *src[5] = {steps_h, tiles_h, level_start_h, wfrag_h, bias_h};
    size_t n[5] = {(size_t)n_steps * 12, (size_t)n_tiles * 2, (size_t)n_levels + 1, (size_t)n_w, (size_t)n_b};
    for (int i = 0; i < 5; i++) {
        HIPCHK(hipMalloc(&c->mnet_bufs[i], n[i] * 4));
        HIPCHK(hipMemcpy(c->mnet_bufs[i], src[i], n[i] * 4, hipMemcpyHostToDevice));
    }
    MfmaNet &m = c->mnet;
    m.steps = (const int32_t *)c->mnet_bufs[0]; m.tiles = (const int32_t *)c->mnet_bufs[1];
    m.level_start = (const int32_t *)c->mnet_bufs[2]; m.wfrag = (const float *)c->mnet_bufs[3];
    m.bias = (const float *)c->mnet_bufs[4];
    m.n_steps = n_steps; m.n_levels = n_levels; m.stride = stride;
    memcpy(m.io, io_h, sizeof m.io);
    return MZ_OK;
}

extern "C" int mz_expand_backup_net(mz_ctx *c, int32_t select_next, const double *noise, int32_t noise_max_calls,
                                    void *stream)
{
    int rc = check_ctx(c, "mz_expand_backup_net");
    if (rc) return rc;
    if (!c->selected) return fail(MZ_ERR_STATE, "mz_expand_backup_net: no pending selection (call mz_select first)");
    if (!c->mnet.steps) return fail(MZ_ERR_STATE, "mz_expand_backup_net: no network uploaded (call mz_set_network_mfma first)");
    int leaf = c->n_expanded + 1;
    size_t lds = (size_t)c->mnet.stride * MZ_NET_TM * 4;
    static bool attr_set = false;
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_net_step), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_net_step, dim3((c->B + MZ_NET_TM - 1) / MZ_NET_TM), dim3(MZ_NET_NT), lds, (hipStream_t)stream, c->d,
                       c->mnet, leaf, c->leaf_logits, c->leaf_value, c->leaf_reward);
    if ((rc = launch_status("mz_expand_backup_net"))) return rc;
    select_next = select_next && leaf < c->cfg.n_simulations;
    if (select_next) {
        if ((rc = check_noise(c, noise, noise_max_calls, "mz_expand_backup_net"))) return rc;
        if (c->cfg.noise_mode != MZ_NOISE_BUFFER) noise = nullptr;
        MZ_LAUNCH(c, k_backup_select, false, c->B, 0, stream, c->d, leaf, (const float *)c->leaf_logits,
                  (const double *)c->leaf_value, (const double *)c->leaf_reward, noise, (int)noise_max_calls);
    } else {
        MZ_LAUNCH(c, k_expand_backup, false, c->B, 0, stream, c->d, leaf, (const float *)c->leaf_logits,
                  (const double *)c->leaf_value, (const double *)c->leaf_reward, (const float *)nullptr);
    }
    c->n_expanded = leaf;
    c->selected = select_next ? 1 : 0;
    return launch_status("mz_expand_backup_net");
}

extern "C" int mz_act_env_step(mz_ctx *c, const mz_env_state *env, double temperature, const double *u,
                               int32_t *out_action, double *out_probs, double *out_root_value,
                               float *out_obs_before, float *out_reward, uint8_t *out_done,
                               int32_t *out_to_play, void *stream)
{
    int rc = check_ctx(c, "mz_act_env_step");
    if (rc) return rc;
    if (c->B < 1) return fail(MZ_ERR_STATE, "mz_act_env_step: no tree");
    if (!env || !out_action || !out_probs || !out_root_value || !out_reward || !out_done || !out_to_play)
        return fail(MZ_ERR_ARG, "mz_act_env_step: null argument");
    if (env->kind != MZ_ENV_TICTACTOE && env->kind != MZ_ENV_CARTPOLE) return fail(MZ_ERR_ARG, "mz_act_env_step: unknown environment kind");
    const int want_A = env->kind == MZ_ENV_TICTACTOE ? 9 : 2, want_obs = env->kind == MZ_ENV_TICTACTOE ? 27 : 4;
    if (c->cfg.n_actions != want_A || env->obs_size != want_obs)
        return fail(MZ_ERR_ARG, "mz_act_env_step: action count / observation size do not match the environment");
    if (!env->state || !env->to_play || !env->legal || !env->cur_obs || !env->search_obs || !env->moves ||
        (env->kind == MZ_ENV_TICTACTOE && !env->mover) || (env->kind == MZ_ENV_CARTPOLE && !env->reset_state))
        return fail(MZ_ERR_ARG, "mz_act_env_step: incomplete environment state");
    EnvDev e;
    e.kind = env->kind; e.max_moves = env->max_moves; e.stale = env->stale_observation; e.obs_size = env->obs_size;
    e.state = env->state; e.mover = env->mover; e.to_play = env->to_play; e.legal = env->legal;
    e.cur_obs = env->cur_obs; e.search_obs = env->search_obs; e.moves = env->moves; e.reset_state = env->reset_state;
    MZ_LAUNCH(c, k_act_env, true, c->B, 0, stream, c->d, e, temperature, u, out_action, out_probs, out_root_value,
              out_obs_before, out_reward, out_done, out_to_play);
    return launch_status("mz_act_env_step");
}
