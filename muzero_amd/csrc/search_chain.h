// Whole searches in ONE launch for networks that are evaluated across workgroups (layers too wide for a tile of trees
// in one CU's LDS, more than 16 actions): the loop of mcts/mcts.py:236-248 with network.py:105-126 as the leaf
// evaluator, for every tree of the batch, without a kernel boundary between its phases.
//
// One workgroup of 8 waves per CU.  Waves 0-3 are NETWORK waves, waves 4-7 are TREE waves.  G consecutive workgroups
// (of one XCD under round-robin placement: speed only) form a CLUSTER that owns tiles of 32 trees.  Per simulation and
// tile the cluster runs the network level by level: a level's column tiles are cut into SEGMENTS (at most 8 column
// tiles of one layer) dealt to the cluster's workgroups on the host; a workgroup stages the 32 x K inputs of a segment
// in LDS and each of its network waves multiplies up to two column tiles, each for both 16-row halves of the tile
// (v_mfma_f32_16x16x4_f32, same k order per output as k_net_level_rows, hence the same results).  The tile's 32 trees
// are served by 32 tree waves of the cluster (one wave per tree: epilogue of the leaf evaluation, expansion, backup,
// the next descent, staging of the next inputs - the code of k_leaf_backup_select).
//
// Everything that crosses workgroups - network inputs, layer outputs - travels as 8-byte GRANULES {value, tag} in
// global memory, written with write-through (sc1) 16-byte stores of two granules and read with sc1 loads that bypass
// the reader's vector cache.  The tag names the simulation and the producing phase, so the data is its own flag: a
// consumer loads what it needs, checks the tags and loads again until all of them match.  No flags, counters, fences
// or drains; a producer never waits.  A region is rewritten (next simulation) only after everything that read it has
// produced data the rewriter waited for, so a tag never runs ahead of a reader.  Every wait is bounded by a clock: a
// wave that waits too long raises the device error flag (3), marks its workgroup dead and leaves; the waves that
// depend on it run into the same bound.
//
// Measured on the Atari-shaped workload (1024 trees x 200 simulations) this kernel is correct - trees bit-identical to
// the level-by-level launches - and SLOWER than they are (12.5-14.7 ms against 8.4 ms): a tile's simulation is a serial
// chain of three hand-offs, each segment pays its staging, a barrier of four waves and 8 KiB of write-through stores
// per wave, and one network wave per SIMD cannot hide any of it.  The engine therefore uses it only on request
// (MZ_SEARCH_CHAIN).
#pragma once
#include "net_mfma.h"

namespace mz {

#ifdef MZ_STAMPS
// diagnostic build: cycles per phase, summed over the search, of network wave 0 (slots 0..7) and tree wave 4 (8..15) of
// every workgroup; 16 slots per workgroup
static __device__ long long mz_chain_stamps[16 * 256];
#define MZ_CSTAMP(k) { long long n_ = clock64(); cst_[k] += n_ - ct_; ct_ = n_; }
#else
#define MZ_CSTAMP(k)
#endif

struct ChainRt {              // per wave
    __amdgpu_buffer_rsrc_t rs;
    const unsigned long long *granules;   // acts2 as granules
    unsigned *sync;           // LDS: [0] arrivals of the network waves' barrier, [1] dead flag
    int32_t *err;
    long long timeout;
    int phases;               // levels + 1
    int stride;
#ifdef MZ_STAMPS
    long long *cst, ct;
#endif
};
#ifdef MZ_STAMPS
#define MZ_RSTAMP(rt, k) { long long n_ = clock64(); (rt).cst[k] += n_ - (rt).ct; (rt).ct = n_; }
#else
#define MZ_RSTAMP(rt, k)
#endif

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
    // (every lane does it, without a branch: a per-lane branch next to the `return` that follows makes the compiler
    // treat the whole enclosing loop as divergent control flow, with its loop-carried scalars in vector registers)
    atomicExch(rt.err, 3);
    __hip_atomic_store(rt.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct ChainClock {
    long long t0;
    __device__ __forceinline__ void start() { t0 = wall_clock64(); }
    // (the clock is one value per wave; said explicitly, or every loop with a bounded wait counts as divergent)
    __device__ __forceinline__ bool expired(const ChainRt &rt) const
    {
        return __builtin_amdgcn_readfirstlane((int)(wall_clock64() - t0 > rt.timeout)) != 0;
    }
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
        if (__builtin_amdgcn_readfirstlane(__all(ok))) return true;
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
        // (one address for the whole wave: the values are uniform, and the compiler must know it, or everything behind
        // this barrier counts as divergent control flow)
        const unsigned n = __builtin_amdgcn_readfirstlane(__hip_atomic_load(rt.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((int)(n - gen) >= 0) break;
        if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(rt.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) return false;
        __builtin_amdgcn_s_sleep(1);
        if (ck.expired(rt)) { chain_give_up(rt); return false; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    return true;
}

// Weights stream through a RING of MZ_CH_RING blocks per column tile: block i of a tile is multiplied out of slot
// i % MZ_CH_RING, and right behind its matrix instructions the slot is asked to fetch block i + MZ_CH_RING.  Every
// block issues the same number of loads (a block the tile does not have is addressed out of range: the hardware
// returns zero and touches no memory), so the loads in flight at any point are a compile-time fact.  A wave keeps two
// rings, one per column tile of its pair: while one tile multiplies, the other ring already holds the first blocks of
// the tile that comes next (this segment's second tile, or the next segment's first).
struct ChainRing { float4 w[MZ_CH_RING]; };

__device__ __forceinline__ float4 chain_wblock(const NetBufs &nb, int w_off, int nkb, int k, int lane)
{
    // the block's place goes into the scalar offset (which takes no part in the range check); the lane's offset is its
    // own 16 bytes, or out of range
    return as_float4(__builtin_amdgcn_raw_buffer_load_b128(nb.w, k < nkb ? lane * 16 : 0x7ffffff0, w_off + k * 1024, 0));
}

__device__ __forceinline__ void chain_prime(const NetBufs &nb, int w_off, int nkb, int lane, ChainRing &ring)
{
#pragma unroll
    for (int i = 0; i < MZ_CH_RING; i++) ring.w[i] = chain_wblock(nb, w_off, nkb, i, lane);
}

// One column tile over the staged inputs of the tile's 32 trees: both 16-row halves share every weight block and
// alternate on the matrix pipe, which hides the dependent latency of a single accumulator.  Ring primed by the caller.
// `kbs`: the 16-input blocks that the weight blocks multiply (scalars).  Per output the k order of mfma_blocks.
template <int K>
__device__ __forceinline__ bool chain_block(const NetBufs &nb, int w_off, int n, int i0, const float *a0, const float *a1,
                                            const int32_t *kx, int lane, ChainRing &ring, f32x4 &acc0, f32x4 &acc1, float4 &x0,
                                            float4 &x1)
{
    // block i0 + K of the tile out of ring slot K; false: the tile has no such block
    const int i = i0 + K;
    if (i >= n) return false;
    // the next block's operands, read unconditionally (the last block reads its own again): a conditional read makes
    // the compiler copy the registers at the join, in front of this block's matrix instructions, and the wave then
    // waits for the LDS round trip there
    const int inext = i + 1 < n ? i + 1 : i;
    const int kb = kx ? __builtin_amdgcn_readfirstlane(kx[inext]) : inext;
    const float4 n0 = *reinterpret_cast<const float4 *>(a0 + kb * 16), n1 = *reinterpret_cast<const float4 *>(a1 + kb * 16);
    __builtin_amdgcn_sched_barrier(0);  // the reads of the next block stay ahead of this block's matrix instructions
    const float4 w = ring.w[K];
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, w.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, w.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, w.y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, w.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.z, w.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, w.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.w, w.w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, w.w, acc1, 0, 0, 0);
    ring.w[K] = chain_wblock(nb, w_off, n, i + MZ_CH_RING, lane);
    x0 = n0; x1 = n1;
    return true;
}

// One column tile over the staged inputs of the tile's 32 trees: both 16-row halves share every weight block and
// alternate on the matrix pipe, which hides the dependent latency of a single accumulator.  Ring primed by the caller.
// `kx`: the 16-input blocks that the weight blocks multiply, or NULL for blocks 0, 1, 2, ...  Per output the k order of
// mfma_blocks.  The blocks go in groups of MZ_CH_RING (slots are register names, so a group is straight-line code; the
// groups are a loop, which keeps the code small enough for the instruction cache beside the tree waves' code).
__device__ __forceinline__ void chain_tile_mfma(const MfmaNet &net, const NetBufs &nb, int w_off, int n, int dense, int kxo,
                                                const float *a0, const float *a1, int lane, ChainRing &ring, f32x4 &acc0,
                                                f32x4 &acc1)
{
    static_assert(MZ_CH_RING == 6, "chain_tile_mfma spells out six ring slots");
    const int32_t *kx = dense ? nullptr : net.kidx + kxo;
    const int kb0 = kx ? __builtin_amdgcn_readfirstlane(kx[0]) : 0;
    float4 x0 = *reinterpret_cast<const float4 *>(a0 + kb0 * 16), x1 = *reinterpret_cast<const float4 *>(a1 + kb0 * 16);
    for (int i0 = 0; i0 < n; i0 += MZ_CH_RING) {
        if (!chain_block<0>(nb, w_off, n, i0, a0, a1, kx, lane, ring, acc0, acc1, x0, x1)) break;
        if (!chain_block<1>(nb, w_off, n, i0, a0, a1, kx, lane, ring, acc0, acc1, x0, x1)) break;
        if (!chain_block<2>(nb, w_off, n, i0, a0, a1, kx, lane, ring, acc0, acc1, x0, x1)) break;
        if (!chain_block<3>(nb, w_off, n, i0, a0, a1, kx, lane, ring, acc0, acc1, x0, x1)) break;
        if (!chain_block<4>(nb, w_off, n, i0, a0, a1, kx, lane, ring, acc0, acc1, x0, x1)) break;
        if (!chain_block<5>(nb, w_off, n, i0, a0, a1, kx, lane, ring, acc0, acc1, x0, x1)) break;
    }
}

// Residual inputs of four consecutive rows of a column (one granule per value, written by an earlier level of this
// simulation) added to v.  Board networks only.
__device__ __forceinline__ bool chain_residual(const ChainRt &rt, size_t idx0, unsigned tag, float (&v)[4])
{
    ChainClock ck;
    ck.start();
    unsigned long long g[4];
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            g[j] = __hip_atomic_load(rt.granules + idx0 + (size_t)j * rt.stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && (unsigned)(g[j] >> 32) == tag;
        }
        if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
        __builtin_amdgcn_s_sleep(2);
        if (ck.expired(rt)) { chain_give_up(rt); return false; }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] += __uint_as_float((unsigned)g[j]);
    return true;
}

// Bias, residual input and activation of one 16 x 16 accumulator, into the wave's patch ([32][MZ_CH_PW] floats: rows
// 16 * rh .., columns 16 * half ..).  C/D layout of 16x16x4: column = lane & 15, row = (lane >> 4) * 4 + register.
__device__ __forceinline__ bool chain_tile_out(const ChainRt &rt, const int32_t *q, int tile, int half, int rh, const f32x4 &acc,
                                               float b, float *patch, int t0, int sim, int lane)
{
    const int c = lane & 15, r4 = rh * 16 + (lane >> 4) * 4;
    const int skip_off = __builtin_amdgcn_readfirstlane(q[3]), actf = __builtin_amdgcn_readfirstlane(q[4]);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = acc[j] + b;
    if (skip_off >= 0 && !chain_residual(rt, (size_t)(t0 + r4) * rt.stride + skip_off + tile * 16 + c,
                                         chain_tag(rt, sim, __builtin_amdgcn_readfirstlane(q[9])), v))
        return false;
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

// One segment as a network wave needs it: scalars out of the workgroup's LDS copy of its descriptors.
struct ChainSeg {
    const int32_t *q;   // descriptor (LDS)
    int KK, in_off, n, dense, wA, wB, kxA, kxB, nkbA, nkbB, tA;
    bool hasA, hasB;
    int t0, sim;        // first tree of the tile, simulation
    unsigned in_tag;
};

__device__ __forceinline__ ChainSeg chain_seg_read(const ChainRt &rt, const int32_t *q, int t0, int sim, int nwave)
{
    auto qs = [&](int i) { return __builtin_amdgcn_readfirstlane(q[i]); };
    ChainSeg g;
    g.q = q; g.t0 = t0; g.sim = sim;
    g.KK = qs(0); g.in_off = qs(1); g.n = qs(6); g.dense = qs(7);
    g.tA = 2 * nwave;
    g.hasA = g.tA < g.n; g.hasB = g.tA + 1 < g.n;
    const int ia = 12 + 3 * (g.hasA ? g.tA : 0), ib = 12 + 3 * (g.hasB ? g.tA + 1 : 0);
    g.wA = qs(ia); g.wB = qs(ib); g.kxA = qs(ia + 2); g.kxB = qs(ib + 2);
    g.nkbA = g.hasA ? qs(ia + 1) : 0; g.nkbB = g.hasB ? qs(ib + 1) : 0;
    g.in_tag = chain_tag(rt, sim, qs(8));
    return g;
}

// A thread's share of a segment's 32 x K inputs: pairs nt, nt + 256, ... of the 16 * K (nt: thread of the 256).  Asked
// for ahead of time, before the previous segment's outputs are stored (vector-memory operations complete in order: a
// load behind a write-through store would wait for that store's round trip).
__device__ __forceinline__ bool chain_fetch_fits(const ChainSeg &g) { return g.KK <= 16 * MZ_CH_NF; }

// (row of pair i = i / pairs per row, by a multiplication: exact for the i < 2^15 that occur)
__device__ __forceinline__ int chain_row_of(int i, unsigned inv) { return (int)__umulhi((unsigned)i, inv); }
__device__ __forceinline__ unsigned chain_inv(int ppr) { return (unsigned)(0x100000000ull / (unsigned)ppr) + 1u; }

__device__ __forceinline__ void chain_issue_fetch(const ChainRt &rt, const ChainSeg &g, int nt, u32x4 (&v)[MZ_CH_NF])
{
    const int ppr = g.KK / 2, total = MZ_CH_TM * ppr;
    const unsigned inv = chain_inv(ppr);
#pragma unroll
    for (int k = 0; k < MZ_CH_NF; k++) {
        const int i = k * 256 + nt;
        const int row = chain_row_of(i, inv), p = i - row * ppr;
        v[k] = gr_load(rt, i < total ? gr_off(rt, g.t0 + row, g.in_off + 2 * p) : 0xfffffff0u);
    }
}

// ... and into the staging buffer once every tag is right; while they are not, the wave looks at a few granules at a
// time (one per row) and fetches again when those have arrived.
__device__ __forceinline__ bool chain_land(ChainRt &rt, const ChainSeg &g, int nt, int lane, u32x4 (&v)[MZ_CH_NF],
                                           bool ahead, float *sb, int ks)
{
    const int ppr = g.KK / 2, total = MZ_CH_TM * ppr;
    const unsigned inv = chain_inv(ppr);
    if (ahead) {
        ChainClock ck;
        ck.start();
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < MZ_CH_NF; k++) ok = ok && (k * 256 + nt >= total || (v[k][1] == g.in_tag && v[k][3] == g.in_tag));
            if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
            {   // wait for the last granule pair of row (lane & 31) in two of the input's 16-float blocks, then ask again
                const int blk = (lane >> 5) * (g.KK / 16) / 2;
                const unsigned off[1] = {gr_off(rt, g.t0 + (lane & 31), g.in_off + blk * 16 + 14)};
                const bool has[1] = {true};
                u32x4 look[1];
                if (!gr_fetch<1, 4>(rt, off, has, g.in_tag, look)) return false;
            }
            chain_issue_fetch(rt, g, nt, v);
            if (ck.expired(rt)) { chain_give_up(rt); return false; }
        }
        MZ_RSTAMP(rt, 0)
#pragma unroll
        for (int k = 0; k < MZ_CH_NF; k++) {
            const int i = k * 256 + nt;
            const int row = chain_row_of(i, inv), p = i - row * ppr;
            if (i < total) *reinterpret_cast<float2 *>(sb + row * ks + 2 * p) = make_float2(__uint_as_float(v[k][0]), __uint_as_float(v[k][2]));
        }
        MZ_RSTAMP(rt, 1)
        return true;
    }
    {   // inputs too wide to fetch ahead: wait for the last granule pair of row (lane & 31) in two of the input's
        // 16-float blocks, then take the inputs in passes
        const int blk = (lane >> 5) * (g.KK / 16) / 2;
        const unsigned off[1] = {gr_off(rt, g.t0 + (lane & 31), g.in_off + blk * 16 + 14)};
        const bool has[1] = {true};
        u32x4 look[1];
        if (!gr_fetch<1, 4>(rt, off, has, g.in_tag, look)) return false;
    }
    MZ_RSTAMP(rt, 0)
    for (int base = 0; base < total; base += 256 * 4) {
        unsigned off[4];
        bool has[4];
        int dst[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + k * 256 + nt;
            has[k] = i < total;
            const int row = chain_row_of(i, inv), p = i - row * ppr;
            off[k] = gr_off(rt, g.t0 + row, g.in_off + 2 * p);
            dst[k] = row * ks + 2 * p;
        }
        u32x4 w[4];
        if (!gr_fetch<4, 2>(rt, off, has, g.in_tag, w)) return false;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (has[k]) *reinterpret_cast<float2 *>(sb + dst[k]) = make_float2(__uint_as_float(w[k][0]), __uint_as_float(w[k][2]));
    }
    MZ_RSTAMP(rt, 1)
    return true;
}

// The outputs of a wave's column tiles: through its patch into the tile's rows, as granule pairs - 32 rows x (8 or 16)
// pairs, a row's pairs in consecutive lanes.
__device__ __forceinline__ bool chain_outputs(ChainRt &rt, const ChainSeg &g, const f32x4 (&acc)[4], float biasA, float biasB,
                                              float *patch, int lane)
{
    if (!chain_tile_out(rt, g.q, g.tA, 0, 0, acc[0], biasA, patch, g.t0, g.sim, lane)) return false;
    if (!chain_tile_out(rt, g.q, g.tA, 0, 1, acc[1], biasA, patch, g.t0, g.sim, lane)) return false;
    if (g.hasB) {
        if (!chain_tile_out(rt, g.q, g.tA + 1, 1, 0, acc[2], biasB, patch, g.t0, g.sim, lane)) return false;
        if (!chain_tile_out(rt, g.q, g.tA + 1, 1, 1, acc[3], biasB, patch, g.t0, g.sim, lane)) return false;
    }
    lds_fence();
    const unsigned otag = chain_tag(rt, g.sim, __builtin_amdgcn_readfirstlane(g.q[10]));
    const int out_off = __builtin_amdgcn_readfirstlane(g.q[2]);
    const int ppr = g.hasB ? 16 : 8;
    for (int i = lane; i < MZ_CH_TM * ppr; i += 64) {
        const int row = i / ppr, p = i - row * ppr;
        const float2 v = *reinterpret_cast<const float2 *>(patch + row * MZ_CH_PW + 2 * p);
        gr_store(rt, gr_off(rt, g.t0 + row, out_off + g.tA * 16 + 2 * p), v.x, v.y, otag);
    }
    lds_fence();
    return true;
}

// The four outputs of a tree's row into the same places of the wave's LDS row.  The pairs of all four are numbered
// through (output k: pairs cnt[k-1] .. cnt[k] - 1, each output rounded up to whole loads of 64 pairs) and taken
// MZ_CH_NS loads per lane at a time: one round trip for the usual sizes.  While a tag is missing the wave looks at the
// last granule pair of every output and asks again when those have arrived.
#define MZ_CH_NS 8
__device__ __forceinline__ bool chain_sweep_outputs(ChainRt &rt, int t, const int (&f0)[4], const int (&n)[4],
                                                    const unsigned (&tag)[4], float *row, int lane)
{
    const int c0 = (n[0] / 2 + 63) / 64, c1 = c0 + (n[1] / 2 + 63) / 64, c2 = c1 + (n[2] / 2 + 63) / 64, c3 = c2 + (n[3] / 2 + 63) / 64;
    bool looked = false;
    for (int base = 0; base < c3; base += MZ_CH_NS) {
        unsigned off[MZ_CH_NS], want[MZ_CH_NS];
        bool has[MZ_CH_NS];
        int dst[MZ_CH_NS];
#pragma unroll
        for (int u = 0; u < MZ_CH_NS; u++) {
            const int i = base + u;
            const int k = i < c0 ? 0 : i < c1 ? 1 : i < c2 ? 2 : 3;
            const int first = k == 0 ? 0 : k == 1 ? c0 : k == 2 ? c1 : c2;
            const int p = (i - first) * 64 + lane;
            has[u] = i < c3 && p < n[k] / 2;
            dst[u] = f0[k] + 2 * p;
            off[u] = gr_off(rt, t, dst[u]);
            want[u] = tag[k];
        }
        u32x4 g[MZ_CH_NS];
        ChainClock ck;
        ck.start();
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < MZ_CH_NS; u++) g[u] = gr_load(rt, has[u] ? off[u] : 0xfffffff0u);
#pragma unroll
            for (int u = 0; u < MZ_CH_NS; u++) ok = ok && (!has[u] || (g[u][1] == want[u] && g[u][3] == want[u]));
            if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
            for (int k = 0; k < 4; k++) {
                const unsigned lo[1] = {gr_off(rt, t, f0[k] + n[k] - 2)};
                const bool lh[1] = {lane == 0};
                u32x4 look[1];
                if (!gr_fetch<1, 8>(rt, lo, lh, tag[k], look)) return false;
            }
            if (ck.expired(rt)) { chain_give_up(rt); return false; }
        }
        if (!looked) { MZ_RSTAMP(rt, 0) looked = true; }
#pragma unroll
        for (int u = 0; u < MZ_CH_NS; u++)
            if (has[u]) *reinterpret_cast<float2 *>(row + dst[u]) = make_float2(__uint_as_float(g[u][0]), __uint_as_float(g[u][2]));
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
    float *stage = smem;                                  // [2][MZ_CH_TM][ks]
    float *patches = stage + 2 * MZ_CH_TM * ks;           // [4][MZ_CH_TM][MZ_CH_PW]
    float *trows = patches + 4 * MZ_CH_TM * MZ_CH_PW;     // [4][stride]
    double *s_bk = reinterpret_cast<double *>(trows + 4 * net.stride);   // [4][64]
    double *s_ct = s_bk + 4 * 64;                         // [MZ_CT_LDS] (PUCT)
    unsigned *sync = reinterpret_cast<unsigned *>(s_ct + (GUMBEL ? 0 : MZ_CT_LDS));   // [16]
    int32_t *lvl_base = reinterpret_cast<int32_t *>(sync + 16);                       // [MZ_CH_MAXL + 1]
    int32_t *seg_lds = lvl_base + MZ_CH_MAXL + 4;                                     // [MZ_CH_MAXSEG][MZ_CH_SEG]
    if (tid < 4) sync[tid] = 0;
    {   // this workgroup's segments, all levels, one after the other
        int base = 0;
        for (int l = 0; l < net.n_levels; l++) {
            const int r0 = ca.blk_range[l * (G + 1) + m], r1 = ca.blk_range[l * (G + 1) + m + 1];
            if (tid == 0) lvl_base[l] = base;
            for (int i = tid; i < (r1 - r0) * MZ_CH_SEG; i += MZ_CH_NT) seg_lds[base * MZ_CH_SEG + i] = ca.seg[(size_t)r0 * MZ_CH_SEG + i];
            base += r1 - r0;
        }
        if (tid == 0) lvl_base[net.n_levels] = base;
    }
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
#ifdef MZ_STAMPS
    long long cst_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    rt.cst = cst_; rt.ct = clock64();
    auto stamps_out = [&](int base) {
        if (lane == 0 && blockIdx.x < 256)
            for (int k = 0; k < 8; k++) mz_chain_stamps[blockIdx.x * 16 + base + k] = cst_[k];
    };
#endif

    if (wave < 4) {
        // ------------------------------------------------------------------ network waves
        // The workgroup's segments in order (simulation, tile, level, segment), software-pipelined: a segment's inputs
        // and first weight blocks are asked for while the segment before it multiplies, and before that segment's
        // outputs are stored.
        const NetBufs nb = net_bufs(net);
        float *patch = patches + wave * MZ_CH_TM * MZ_CH_PW;
        const int nt = wave * 64 + lane;
        unsigned bar_gen = 0;
        int buf = 0;
        // (everything that steers the loop is wave-uniform and is told so to the compiler: values read from LDS count as
        // per-lane otherwise, and with them every branch and every scalar operand derived from them)
        auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
        int s = 0, j = c, l = 0, r = uni(lvl_base[0]), r1 = uni(lvl_base[1]);   // r, r1: positions in the LDS copy
        bool done = false;
        auto settle = [&]() {   // (s, j, l, r) onto the next existing segment
            while (r >= r1) {
                if (++l == net.n_levels) {
                    l = 0;
                    if ((j += ca.n_clusters) >= ca.n_tiles) { j = c; if (++s == ca.n_sims) { done = true; return; } }
                }
                r = uni(lvl_base[l]); r1 = uni(lvl_base[l + 1]);
            }
        };
        settle();
        if (done) return;
        ChainRing ringA, ringB;
        u32x4 fin[MZ_CH_NF];
        ChainSeg cur = chain_seg_read(rt, seg_lds + MZ_CH_SEG * r, j * MZ_CH_TM, s, wave);
        chain_prime(nb, cur.wA, cur.nkbA, lane, ringA);
        bool ahead = chain_fetch_fits(cur);
        if (ahead) chain_issue_fetch(rt, cur, nt, fin);
        for (;;) {
            float *sb = stage + buf * MZ_CH_TM * ks;
            // biases of this segment's tiles (needed behind the matrix instructions)
            const int b0 = __builtin_amdgcn_readfirstlane(cur.q[5]) + (lane & 15);
            const float biasA = cur.hasA ? net.bias[b0 + cur.tA * 16] : 0.0f, biasB = cur.hasB ? net.bias[b0 + cur.tA * 16 + 16] : 0.0f;
            if (!chain_land(rt, cur, nt, lane, fin, ahead, sb, ks)) return;
            if (!net_barrier(rt, bar_gen)) return;
            MZ_RSTAMP(rt, 2)
            // the segment after this one, known before the matrix work so that its first loads can leave early
            r++;
            settle();
            ChainSeg nxt = cur;
            if (!done) nxt = chain_seg_read(rt, seg_lds + MZ_CH_SEG * r, j * MZ_CH_TM, s, wave);
            f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const float *a0 = sb + (lane & 15) * ks + (lane >> 4) * 4, *a1 = a0 + 16 * ks;
            if (cur.hasB) chain_prime(nb, cur.wB, cur.nkbB, lane, ringB);
            if (cur.hasA) chain_tile_mfma(net, nb, cur.wA, cur.nkbA, cur.dense, cur.kxA, a0, a1, lane, ringA, acc[0], acc[1]);
            if (!done) chain_prime(nb, nxt.wA, nxt.nkbA, lane, ringA);
            if (cur.hasB) chain_tile_mfma(net, nb, cur.wB, cur.nkbB, cur.dense, cur.kxB, a0, a1, lane, ringB, acc[2], acc[3]);
#ifdef MZ_STAMPS
            if (acc[0][0] == 12345.678f && acc[3][1] == 3.0f) rt.ct++;   // the stamp waits for the matrix instructions
#endif
            MZ_RSTAMP(rt, 3)
            if (!done) {
                ahead = chain_fetch_fits(nxt);
                if (ahead) chain_issue_fetch(rt, nxt, nt, fin);
            }
            if (cur.hasA && !chain_outputs(rt, cur, acc, biasA, biasB, patch, lane)) return;
            MZ_RSTAMP(rt, 4)
            if (done) break;
            cur = nxt;
            buf ^= 1;
        }
#ifdef MZ_STAMPS
        if (wave == 0) stamps_out(0);
#endif
        return;
    }

    // ---------------------------------------------------------------------- tree waves
    const int w = wave - 4, slot = m * 4 + w, grp = slot / MZ_CH_TM, r = slot % MZ_CH_TM, n_grp = 4 * G / MZ_CH_TM;
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
                unsigned otag[4];
#pragma unroll
                for (int k = 0; k < 4; k++) otag[k] = chain_tag(rt, s - 1, ca.out_phase[k]);
                if (!chain_sweep_outputs(rt, t, out_f0, out_n, otag, row, lane)) return;
                lds_fence();
                MZ_RSTAMP(rt, 1)
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
#ifdef MZ_STAMPS
                    if (value == 12345.678 && lg == 3.0f) rt.ct++;
#endif
                    MZ_RSTAMP(rt, 2)
                    expand_backup<64, GUMBEL>(d, mem, a, on, parent, action, ntp, depth, leaf, lg, value, reward, bk);
                    if (a == 0) d.visit_total[t] += depth + 1;
                    __threadfence_block();
                }
                MZ_RSTAMP(rt, 3)
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
                    MZ_RSTAMP(rt, 4)
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
                MZ_RSTAMP(rt, 5)
            } else if (live && a == 0) {
                d.sel_depth[t] = -1;
            }
        }
#ifdef MZ_STAMPS
    if (wave == 4) stamps_out(8);
#endif
}

}  // namespace mz
