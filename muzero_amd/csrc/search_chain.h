// Whole searches in ONE launch for networks that are evaluated across workgroups (layers too wide for a tile of trees
// in one CU's LDS, more than 16 actions): the loop of mcts/mcts.py:236-248 with network.py:105-126 as the leaf
// evaluator, for every tree of the batch, without a kernel boundary between its phases.
//
// One workgroup of 8 waves per CU.  Waves 0-3 are NETWORK waves, waves 4-7 are TREE waves.  G consecutive workgroups
// (of one XCD under round-robin placement: speed only) form a CLUSTER that owns tiles of 16 trees.  Per simulation and
// tile the cluster runs the network level by level: a level's column tiles are cut into SEGMENTS (at most 8 column
// tiles of one layer) dealt to the cluster's workgroups on the host; a workgroup stages the 16 x K inputs of a segment
// in LDS and each of its network waves multiplies two adjacent column tiles (v_mfma_f32_16x16x4_f32, same k order per
// output as k_net_level_rows, hence the same results).  The tile's 16 trees are served by 16 tree waves of the cluster
// (one wave per tree: epilogue of the leaf evaluation, expansion, backup, the next descent, staging of the next
// inputs - the code of k_leaf_backup_select).
//
// Everything that crosses workgroups - network inputs, layer outputs - travels as 8-byte GRANULES {value, tag} in
// global memory, written with write-through (sc1) 16-byte stores of two granules and read with sc1 loads that bypass
// the reader's vector cache.  The tag names the simulation and the producing phase, so the data is its own flag: a
// consumer loads what it needs, checks the tags and loads again until all of them match.  No flags, counters, fences
// or drains; a producer never waits.  A region is rewritten (next simulation) only after everything that read it has
// produced data the rewriter waited for, so a tag never runs ahead of a reader.  Every wait is bounded by a clock: a
// wave that waits too long raises the device error flag (3), marks its workgroup dead and leaves; the waves that
// depend on it run into the same bound.
#pragma once
#include "net_mfma.h"

namespace mz {

struct ChainRt {              // per wave
    __amdgpu_buffer_rsrc_t rs;
    const unsigned long long *granules;   // acts2 as granules
    unsigned *sync;           // LDS: [0] arrivals of the network waves' barrier, [1] dead flag
    int32_t *err;
    long long timeout;
    int phases;               // levels + 1
    int stride;
};

__device__ __forceinline__ unsigned chain_tag(const ChainRt &rt, int sim, int phase)
{
    return (unsigned)(sim * rt.phases + phase + 1);
}

// byte offset of the granule of float `f` in row `t`
__device__ __forceinline__ unsigned gr_off(const ChainRt &rt, int t, int f)
{
    return ((unsigned)t * (unsigned)rt.stride + (unsigned)f) * 8u;
}

__device__ __forceinline__ u32x4 gr_load(const ChainRt &rt, unsigned byte_off)
{
    return __builtin_amdgcn_raw_buffer_load_b128(rt.rs, (int)byte_off, 0, 16);   // sc1
}

__device__ __forceinline__ void gr_store(const ChainRt &rt, unsigned byte_off, float v0, float v1, unsigned tag)
{
    u32x4 g = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
    __builtin_amdgcn_raw_buffer_store_b128(g, rt.rs, (int)byte_off, 0, 16);      // sc1: written through
}

// A wait ran out of time: flag it, release the workgroup's other waves.
__device__ __forceinline__ void chain_give_up(const ChainRt &rt)
{
    if ((threadIdx.x & 63) == 0) {
        atomicExch(rt.err, 3);
        __hip_atomic_store(rt.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

struct ChainClock {
    long long t0;
    __device__ __forceinline__ void start() { t0 = wall_clock64(); }
    __device__ __forceinline__ bool expired(const ChainRt &rt) const { return wall_clock64() - t0 > rt.timeout; }
};

// Loads N granule pairs per lane (the lane's k-th pair at byte offset off[k]; has[k] false: none) until every tag
// equals `tag`.  Returns false when the wait ran out of time.
template <int N, int SLEEP>
__device__ __forceinline__ bool gr_fetch(const ChainRt &rt, const unsigned (&off)[N], const bool (&has)[N], unsigned tag,
                                         u32x4 (&g)[N])
{
    ChainClock ck;
    ck.start();
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < N; k++) g[k] = gr_load(rt, has[k] ? off[k] : 0xfffffff0u);
#pragma unroll
        for (int k = 0; k < N; k++) ok = ok && (!has[k] || (g[k][1] == tag && g[k][3] == tag));
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(SLEEP);
        asm volatile("" ::: "memory");
        if (ck.expired(rt)) { chain_give_up(rt); return false; }
    }
}

// Barrier of the four network waves (the tree waves of the workgroup take no part, so s_barrier cannot serve): a
// counter in LDS; `gen` counts this wave's barriers.  LDS operations of a wave complete in order, so the arrival
// follows the wave's staging stores.
__device__ __forceinline__ bool net_barrier(const ChainRt &rt, unsigned &gen)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    gen += 4;
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(rt.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ChainClock ck;
    ck.start();
    for (;;) {
        const unsigned n = __hip_atomic_load(rt.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((int)(n - gen) >= 0) break;
        if (__hip_atomic_load(rt.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;
        __builtin_amdgcn_s_sleep(1);
        if (ck.expired(rt)) { chain_give_up(rt); return false; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    return true;
}

struct ChainW { float4 w[MZ_CH_KB]; };

// MZ_CH_KB weight blocks of a column tile from block k0 on (a fixed number of loads: blocks it does not have are
// addressed out of range and cost no memory access)
__device__ __forceinline__ void chain_wload(const NetBufs &nb, int w_off, int nkb, int k0, int lane, ChainW &r)
{
    // the block's place goes into the scalar offset (which takes no part in the range check); the lane's offset is its
    // own 16 bytes, or out of range for a block the tile does not have
    const int vin = lane * 16, vout = 0x7ffffff0;
#pragma unroll
    for (int i = 0; i < MZ_CH_KB; i++) {
        const int k = k0 + i;
        r.w[i] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(nb.w, k < nkb ? vin : vout, w_off + k * 1024, 0));
    }
}

// One or two column tiles over the staged inputs of 16 trees: block i of the list multiplies the 16 inputs at kb[i].
// Two tiles share every operand read and alternate on the matrix pipe, which hides the dependent latency of a single
// accumulator; per output the k order is that of mfma_blocks.
template <bool TWO>
__device__ __forceinline__ void chain_mfma(int n, const float *a0, const ChainW &wa, const ChainW &wb,
                                           const int (&kb)[MZ_CH_KB], f32x4 &accA, f32x4 &accB)
{
#pragma unroll
    for (int i = 0; i < MZ_CH_KB; i++) {
        if (i < n) {
            const float4 x = *reinterpret_cast<const float4 *>(a0 + kb[i] * 16);
            accA = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, wa.w[i].x, accA, 0, 0, 0);
            if constexpr (TWO) accB = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, wb.w[i].x, accB, 0, 0, 0);
            accA = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, wa.w[i].y, accA, 0, 0, 0);
            if constexpr (TWO) accB = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, wb.w[i].y, accB, 0, 0, 0);
            accA = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, wa.w[i].z, accA, 0, 0, 0);
            if constexpr (TWO) accB = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, wb.w[i].z, accB, 0, 0, 0);
            accA = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, wa.w[i].w, accA, 0, 0, 0);
            if constexpr (TWO) accB = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, wb.w[i].w, accB, 0, 0, 0);
        }
    }
}

// Bias, residual input and activation of one column tile's accumulator, into the wave's patch ([16][MZ_CH_PW] floats,
// columns 16 * half ..).  C/D layout of 16x16x4: column = lane & 15, row = (lane >> 4) * 4 + register.
__device__ __forceinline__ bool chain_tile_out(const ChainRt &rt, const MfmaNet &net, const int32_t *q, int tile, int half,
                                               const f32x4 &acc, float *patch, int t0, int sim, int lane)
{
    const int c = lane & 15, r4 = (lane >> 4) * 4;
    const int skip_off = __builtin_amdgcn_readfirstlane(q[3]), actf = __builtin_amdgcn_readfirstlane(q[4]);
    const float b = net.bias[__builtin_amdgcn_readfirstlane(q[5]) + tile * 16 + c];
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = acc[j] + b;
    if (skip_off >= 0) {   // residual input: one granule per value, written by an earlier level of this simulation
        const unsigned tag = chain_tag(rt, sim, __builtin_amdgcn_readfirstlane(q[9]));
        ChainClock ck;
        ck.start();
        unsigned long long g[4];
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const size_t idx = (size_t)(t0 + r4 + j) * rt.stride + skip_off + tile * 16 + c;
                g[j] = __hip_atomic_load(rt.granules + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (unsigned)(g[j] >> 32) == tag;
            }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(2);
            if (ck.expired(rt)) { chain_give_up(rt); return false; }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] += __uint_as_float((unsigned)g[j]);
    }
    if (actf == 1) {
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
    } else if (actf == 2) {
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = apply_act(v[j], 2);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) patch[(r4 + j) * MZ_CH_PW + half * 16 + c] = v[j];
    return true;
}

// One segment for the 16 trees from t0 on, by the four network waves of a workgroup.  `sb`: this segment's staging
// buffer ([16][ks] floats), `patch`: the wave's output patch.
__device__ __forceinline__ bool chain_segment(const ChainRt &rt, const MfmaNet &net, const NetBufs &nb, const int32_t *q,
                                              int t0, int sim, float *sb, int ks, float *patch, unsigned &bar_gen,
                                              int nwave, int lane)
{
    // (descriptor fields are uniform; read through the vector path, they are made scalars again here)
    auto qs = [&](int i) { return __builtin_amdgcn_readfirstlane(q[i]); };
    const int KK = qs(0), in_off = qs(1), n = qs(6), dense = qs(7);
    const int tA = 2 * nwave, tB = tA + 1;
    const bool hasA = tA < n, hasB = tB < n;
    const int ia = 12 + 3 * (hasA ? tA : 0), ib = 12 + 3 * (hasB ? tB : 0);
    const int wA = qs(ia), wB = qs(ib), kxA = qs(ia + 2), kxB = qs(ib + 2);
    const int nkbA = hasA ? qs(ia + 1) : 0, nkbB = hasB ? qs(ib + 1) : 0;
    // the weights do not depend on the inputs: they travel while the wave waits for those
    ChainW wa, wb;
    chain_wload(nb, wA, nkbA, 0, lane, wa);
    chain_wload(nb, wB, nkbB, 0, lane, wb);
    const unsigned tag = chain_tag(rt, sim, qs(8));
    {   // a cheap look first: the last granule pair of row (lane & 15) in four of the input's 16-float blocks
        const int blk = (lane >> 4) * (KK / 16) / 4;
        const unsigned off[1] = {gr_off(rt, t0 + (lane & 15), in_off + blk * 16 + 14)};
        const bool has[1] = {true};
        u32x4 g[1];
        if (!gr_fetch<1, 4>(rt, off, has, tag, g)) return false;
    }
    // all 16 x KK inputs into LDS: thread i of the 256 takes pairs i, i + 256, ... of the 8 * KK
    const int nt = nwave * 64 + lane, pairs_per_row = KK / 2, total = 16 * pairs_per_row;
    for (int base = 0; base < total; base += 256 * 4) {
        unsigned off[4];
        bool has[4];
        int dst[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + k * 256 + nt;
            has[k] = i < total;
            const int row = i / pairs_per_row, p = i - row * pairs_per_row;
            off[k] = gr_off(rt, t0 + row, in_off + 2 * p);
            dst[k] = row * ks + 2 * p;
        }
        u32x4 g[4];
        if (!gr_fetch<4, 2>(rt, off, has, tag, g)) return false;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (has[k]) *reinterpret_cast<float2 *>(sb + dst[k]) = make_float2(__uint_as_float(g[k][0]), __uint_as_float(g[k][2]));
    }
    if (!net_barrier(rt, bar_gen)) return false;
    if (!hasA) return true;
    const float *a0 = sb + (lane & 15) * ks + (lane >> 4) * 4;
    f32x4 accA = {0, 0, 0, 0}, accB = {0, 0, 0, 0};
    int kb[MZ_CH_KB];
    if (dense && hasB && nkbA == nkbB) {   // two tiles side by side
        for (int k0 = 0; k0 < nkbA; k0 += MZ_CH_KB) {
            if (k0) { chain_wload(nb, wA, nkbA, k0, lane, wa); chain_wload(nb, wB, nkbB, k0, lane, wb); }
#pragma unroll
            for (int i = 0; i < MZ_CH_KB; i++) kb[i] = k0 + i;
            chain_mfma<true>(nkbA - k0 < MZ_CH_KB ? nkbA - k0 : MZ_CH_KB, a0, wa, wb, kb, accA, accB);
        }
    } else {                              // block-sparse or unequal tiles: one after the other
        for (int k0 = 0; k0 < nkbA; k0 += MZ_CH_KB) {
            if (k0) chain_wload(nb, wA, nkbA, k0, lane, wa);
#pragma unroll
            for (int i = 0; i < MZ_CH_KB; i++) kb[i] = dense ? k0 + i : (k0 + i < nkbA ? net.kidx[kxA + k0 + i] : 0);
            chain_mfma<false>(nkbA - k0 < MZ_CH_KB ? nkbA - k0 : MZ_CH_KB, a0, wa, wa, kb, accA, accA);
        }
        for (int k0 = 0; k0 < nkbB; k0 += MZ_CH_KB) {
            if (k0) chain_wload(nb, wB, nkbB, k0, lane, wb);
#pragma unroll
            for (int i = 0; i < MZ_CH_KB; i++) kb[i] = dense ? k0 + i : (k0 + i < nkbB ? net.kidx[kxB + k0 + i] : 0);
            chain_mfma<false>(nkbB - k0 < MZ_CH_KB ? nkbB - k0 : MZ_CH_KB, a0, wb, wb, kb, accB, accB);
        }
    }
    if (!chain_tile_out(rt, net, q, tA, 0, accA, patch, t0, sim, lane)) return false;
    if (hasB && !chain_tile_out(rt, net, q, tB, 1, accB, patch, t0, sim, lane)) return false;
    lds_fence();
    // the patch leaves as granule pairs: 16 rows x (8 or 16) pairs, a row's pairs in consecutive lanes
    const unsigned otag = chain_tag(rt, sim, qs(10));
    const int out_off = qs(2);
    const int ppr = hasB ? 16 : 8;
    for (int i = lane; i < 16 * ppr; i += 64) {
        const int row = i / ppr, p = i - row * ppr;
        const float2 v = *reinterpret_cast<const float2 *>(patch + row * MZ_CH_PW + 2 * p);
        gr_store(rt, gr_off(rt, t0 + row, out_off + tA * 16 + 2 * p), v.x, v.y, otag);
    }
    lds_fence();
    return true;
}

// `n` floats (a multiple of 16) of a tree's row, from float `f0` on, into the same place of the wave's LDS row.
__device__ __forceinline__ bool chain_sweep(const ChainRt &rt, int t, int f0, int n, unsigned tag, float *row, int lane)
{
    const int pairs = n / 2;
    for (int base = 0; base < pairs; base += 64 * 5) {
        unsigned off[5];
        bool has[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int p = base + k * 64 + lane;
            has[k] = p < pairs;
            off[k] = gr_off(rt, t, f0 + 2 * p);
        }
        u32x4 g[5];
        if (!gr_fetch<5, 2>(rt, off, has, tag, g)) return false;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int p = base + k * 64 + lane;
            if (has[k]) *reinterpret_cast<float2 *>(row + f0 + 2 * p) = make_float2(__uint_as_float(g[k][0]), __uint_as_float(g[k][2]));
        }
    }
    return true;
}

template <int RL, bool GUMBEL>
__global__ __launch_bounds__(MZ_CH_NT) void k_search_chain(Dev d, MfmaNet net, ChainArgs ca)
{
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = ca.G;
    int c, m;   // cluster and place in it; workgroups b and b + 8 share an XCD under round-robin placement
    if (gridDim.x % (8 * G) == 0) { const int x = blockIdx.x & 7, i = blockIdx.x >> 3; c = (i / G) * 8 + x; m = i % G; }
    else { c = blockIdx.x / G; m = blockIdx.x % G; }
    const int ks = ca.kk_max + 4;
    float *stage = smem;                                  // [2][16][ks]
    float *patches = stage + 2 * 16 * ks;                 // [4][16][MZ_CH_PW]
    float *trows = patches + 4 * 16 * MZ_CH_PW;           // [4][stride]
    double *s_bk = reinterpret_cast<double *>(trows + 4 * net.stride);   // [4][64]
    double *s_ct = s_bk + 4 * 64;                         // [MZ_CT_LDS] (PUCT)
    unsigned *sync = reinterpret_cast<unsigned *>(s_ct + (GUMBEL ? 0 : MZ_CT_LDS));
    if (tid < 4) sync[tid] = 0;
    for (int i = tid; i < 4 * net.stride; i += MZ_CH_NT) trows[i] = 0.0f;
    const double *ct = nullptr;
    if constexpr (!GUMBEL) {
        if (d.L.S + 2 <= MZ_CT_LDS) {
            for (int i = tid; i < d.L.S + 2; i += MZ_CH_NT) s_ct[i] = d.c_table[i];
            ct = s_ct;
        }
    }
    __syncthreads();
    if (c >= ca.n_clusters || c >= ca.n_tiles) return;
    ChainRt rt;
    rt.rs = __builtin_amdgcn_make_buffer_rsrc(ca.acts2, 0, ca.acts2_bytes, 0x00020000);
    rt.granules = reinterpret_cast<const unsigned long long *>(ca.acts2);
    rt.sync = sync; rt.err = d.err; rt.timeout = ca.timeout; rt.phases = net.n_levels + 1; rt.stride = net.stride;

    if (wave < 4) {
        // ------------------------------------------------------------------ network waves
        const NetBufs nb = net_bufs(net);
        float *patch = patches + wave * 16 * MZ_CH_PW;
        unsigned bar_gen = 0;
        int buf = 0;
        for (int s = 0; s < ca.n_sims; s++)
            for (int j = c; j < ca.n_tiles; j += ca.n_clusters)
                for (int l = 0; l < net.n_levels; l++) {
                    const int r0 = ca.blk_range[l * (G + 1) + m], r1 = ca.blk_range[l * (G + 1) + m + 1];
                    for (int r = r0; r < r1; r++) {
                        if (!chain_segment(rt, net, nb, ca.seg + (size_t)MZ_CH_SEG * r, j * MZ_CH_TM, s, stage + buf * 16 * ks, ks,
                                           patch, bar_gen, wave, lane))
                            return;
                        buf ^= 1;
                    }
                }
        return;
    }

    // ---------------------------------------------------------------------- tree waves
    const int w = wave - 4, slot = m * 4 + w, grp = slot >> 4, r = slot & 15, n_grp = G >> 2;
    float *row = trows + w * net.stride;
    double *bk = s_bk + w * 64;
    const int a = lane, H = net.io[6];
    const bool on = a < d.L.A;
    const bool wide = net.io[9] > 64 && net.io[9] <= 640 && net.io[10] == 1 && H > 16 && H <= 256 && !net.plane_major;
    const int out_f0[4] = {net.io[2], net.io[3], net.io[4], net.io[5]};
    const int out_n[4] = {(H + 15) & ~15, (net.io[9] + 15) & ~15, (net.io[7] + 15) & ~15, (net.io[9] + 15) & ~15};
    for (int s = 0; s <= ca.n_sims; s++)
        for (int jl = grp;; jl += n_grp) {
            const int j = c + jl * ca.n_clusters;
            if (j >= ca.n_tiles) break;
            const int t = j * MZ_CH_TM + r;
            const bool live = t < d.B;
            const int tt = live ? t : 0;
            const TreeMem mem = tree_mem(d, tt);
            float *hid = d.hidden + (long long)tt * d.L.N * H;
            if (s > 0) {   // the outputs of simulation s - 1: epilogue, expansion, backup
                const int leaf = ca.first_leaf + s - 1;
                int parent = 0, action = 0, ntp = 0, depth = 0;
                if (live) { parent = d.sel_parent[t]; action = d.sel_action[t]; ntp = d.sel_ntp[t]; depth = d.sel_depth[t]; }
                // a cheap look first: the last granule pair of each of the four outputs (they may come from different levels)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned off[1] = {gr_off(rt, t, out_f0[k] + out_n[k] - 2)};
                    const bool has[1] = {lane == 0};
                    u32x4 g[1];
                    if (!gr_fetch<1, 8>(rt, off, has, chain_tag(rt, s - 1, ca.out_phase[k]), g)) return;
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (!chain_sweep(rt, t, out_f0[k], out_n[k], chain_tag(rt, s - 1, ca.out_phase[k]), row, lane)) return;
                lds_fence();
                if (live) {
                    if (depth < 1 || depth > d.L.S || parent < 0 || parent >= leaf) {
                        if (a == 0) atomicExch(d.err, 1);
                        chain_give_up(rt);
                        return;
                    }
                    double value, reward;
                    const float lg = on ? row[net.io[4] + a] : 0.0f;
                    if (wide) {
                        LeafRegs<10, 4> lr;
                        leaf_epilogue_load<10, 4>(net, row, a, lr);
                        leaf_epilogue_regs<10, 4>(net, lr, a, hid + (long long)leaf * H, &value, &reward);
                    } else if (net.io[9] <= 64) {
                        tile_epilogue<64, 16>(net, row, a, hid + (long long)leaf * H, &value, &reward);
                    } else {
                        tile_epilogue<64, 64>(net, row, a, hid + (long long)leaf * H, &value, &reward);
                    }
                    expand_backup<64, GUMBEL>(d, mem, a, on, parent, action, ntp, depth, leaf, lg, value, reward, bk);
                    if (a == 0) d.visit_total[t] += depth + 1;
                    __threadfence_block();
                }
            }
            if (s < ca.n_sims) {   // the next descent and its network inputs
                if (live) {
                    const double *nz = ca.noise ? ca.noise + (long long)t * ca.noise_max_calls * d.L.A : nullptr;
                    DescendCtx dc = descend_ctx<GUMBEL>(d, t, a, on, ct, nullptr);
                    uint32_t calls = GUMBEL ? 0u : d.noise_calls[t];
                    int action, ntp, depth;
                    const int parent = descend<64, RL, GUMBEL>(d, dc, mem, a, on, nz, ca.noise_max_calls, calls, &action, &ntp, &depth);
                    if (a == 0) {
                        if constexpr (!GUMBEL) store_noise_calls(d, t, calls, ca.noise_max_calls);
                        d.sel_parent[t] = parent; d.sel_action[t] = action; d.sel_ntp[t] = ntp; d.sel_depth[t] = depth;
                    }
                    tile_stage_inputs<64>(net, row, hid + (long long)parent * H, action, a);
                }
                // what follows the inputs up to the next multiple of 16 floats is multiplied by zero weights and must be
                // finite; rows past the end of the batch feed zeros
                for (int f = (live ? H + net.io[8] : 0) + a; f < ca.in_n; f += 64) row[ca.in_lo + f] = 0.0f;
                lds_fence();
                const unsigned tag = chain_tag(rt, s, 0);
                for (int p = lane; p < ca.in_n / 2; p += 64) {
                    const float2 v = *reinterpret_cast<const float2 *>(row + ca.in_lo + 2 * p);
                    gr_store(rt, gr_off(rt, t, ca.in_lo + 2 * p), v.x, v.y, tag);
                }
                lds_fence();
            } else if (live && a == 0) {
                d.sel_depth[t] = -1;
            }
        }
}

}  // namespace mz
