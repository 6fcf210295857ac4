// Batched leaf evaluation on the matrix cores for networks whose weights do not fit in LDS.
//
// One workgroup (4 waves) evaluates the lowered network (dense program, MFMA layout of
// muzero_amd/packing.py) for a tile of TM = 32 trees.  The tile's activations live in LDS
// ([32][stride] floats, stride = 4 x odd so the 16 rows of an operand read start in 16 different bank groups); weights
// stream from L2 in v_mfma_f32_16x16x4_f32 operand order, one 16-byte load per lane per four k-steps, and every loaded fragment is used for both 16-row halves
// of the tile (two independent accumulators, which also covers the MFMA's dependent latency).  Weights are block-sparse:
// a (step, column tile) job lists the 16-input blocks that hold a non-zero weight and only those are loaded and multiplied.
// Within a level the (step, column tile) jobs are dealt round-robin to the four waves; levels are
// separated by a workgroup barrier.  float32 in, float32 accumulate: a k-ordered fma chain per
// output, same rounding class as the reference's float32 network.
#pragma once
#include "dense_net.h"
#include "mfma_launch.h"

namespace mz {

#ifdef MZ_STAMPS
static __device__ long long mz_dbg_stamps[8 * 1024];   // diagnostic build: cycles per phase, 8 slots per block
#define MZ_KSTAMP(k) { long long n_ = clock64(); if (tid == 0 && blockIdx.x < 1024) mz_dbg_stamps[blockIdx.x * 8 + (k)] = n_ - kst_; kst_ = n_; }
// the same for kernels that name their slot themselves (slot < 0: none)
#define MZ_SSTAMP(slot, k) { long long n_ = clock64(); if ((slot) >= 0) mz_dbg_stamps[(slot) * 8 + (k)] = n_ - kst_; kst_ = n_; }
#else
#define MZ_KSTAMP(k)
#define MZ_SSTAMP(slot, k)
#endif

#define MZ_NET_TM 32
#define MZ_NET_NT 512  // 8 waves: two per SIMD, so one wave's weight-load and LDS waits hide behind the other's MFMAs


using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float apply_act(float v, int act)
{
    if (act == 1) return v > 0.0f ? v : 0.0f;
    if (act == 2) return v > 0.0f ? v : expm1f(v);
    return v;
}

// Operand layout of a (step, column tile) job.  Lane (r = lane & 15, q = lane >> 4) supplies, for the j-th of four
// k-steps of a 16-wide k block, X[tree r][k = 16*blk + 4q + j] (four consecutive floats of its tree's activations: one
// 16-byte LDS read) and the matching weight W[col r][same k] (one 16-byte global read of the pre-swizzled fragment).
// The WEIGHTS are the matrix instruction's first operand and the activations its second: the result then has output
// columns down the registers - lane (r, q) holds columns 4q .. 4q + 3 of tree r - so a job's outputs, its residual
// input and its biases are single 16-byte accesses per lane (with the operands the other way round a lane held one
// column of four trees: four scattered 4-byte accesses each).
#define MZ_WPF 10  // weight blocks fetched per round (k blocks of 16; a job with more runs in rounds of this)
#ifndef MZ_TILE_ASK
#define MZ_TILE_ASK 2  // weight blocks of the NEXT job that a tile-resident job asks for behind each of its own blocks
#endif

// ---------------------------------------------------------------------------------------------------
// Tile-resident evaluation (activations of the block's trees in LDS), software-pipelined per wave.
// The wave's jobs come from the host-made schedule (muzero_amd/packing.py).  Their descriptors are read once per
// kernel into registers - lane j keeps job j - and fetched with lane reads, so a job costs no table loads.  While a
// job's matrix instructions run, the weights of the wave's NEXT job (of this level or a later one: weights do not
// depend on activations) are already on their way: every job issues exactly MZ_WPF weight loads and one bias load
// through buffer descriptors, blocks it does not have are addressed out of range (the hardware returns zero without
// touching memory), so the number of loads in flight is a compile-time fact and the wait for the current job's
// weights never waits for the next job's.  Inside a job the LDS reads of block i+1 are issued before the matrix
// instructions of block i.
struct TileSched {
    int f[MZ_DESC_INTS];  // lane j: descriptor of the wave's j-th job
    int lvl_end;          // lane l: number of the wave's jobs in levels 0..l
    int n_jobs;           // wave-uniform
};

struct WReg {
    float4 w[MZ_WPF];
    float4 bias;   // of output columns 4q .. 4q + 3 of the job's tile (q = lane >> 4)
};

typedef unsigned int u32x4 __attribute__((vector_size(16)));

// (element reads go through scalars: __builtin_bit_cast applied to a vector element directly reads element 0 with this compiler)
__device__ __forceinline__ float4 as_float4(u32x4 v)
{
    const unsigned a = v[0], b = v[1], c = v[2], d = v[3];
    return make_float4(__uint_as_float(a), __uint_as_float(b), __uint_as_float(c), __uint_as_float(d));
}

struct NetBufs {
    __amdgpu_buffer_rsrc_t w, b;
    const char *wbase;   // the weight fragments, for descriptors of a single job's blocks
};

__device__ __forceinline__ NetBufs net_bufs(const MfmaNet &net)
{
    NetBufs r;
    r.w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(net.wfrag), 0, net.w_bytes, 0x00020000);
    r.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(net.bias), 0, net.b_bytes, 0x00020000);
    r.wbase = reinterpret_cast<const char *>(net.wfrag);
    return r;
}

template <int TM>
__device__ __forceinline__ void tile_sched_load(const MfmaNet &net, int wave_v, int lane, TileSched &ts)
{
    const int wave = __builtin_amdgcn_readfirstlane(wave_v);
    const int32_t *ws = (TM == 32 ? net.wstart8 : net.wstart4) + wave * (net.n_levels + 1);
    const int32_t *wd = TM == 32 ? net.wdesc8 : net.wdesc4;
    const int base = ws[0];
    ts.n_jobs = ws[net.n_levels] - base;
#pragma unroll
    for (int k = 0; k < MZ_DESC_INTS; k++) ts.f[k] = lane < ts.n_jobs ? wd[(base + lane) * MZ_DESC_INTS + k] : 0;
    ts.lvl_end = lane < net.n_levels ? ws[lane + 1] - base : ts.n_jobs;
}

__device__ __forceinline__ int sched_get(const TileSched &ts, int field, int job)
{
    return __builtin_amdgcn_readlane(ts.f[field], job);
}

// Issues the loads of job `job` (exactly MZ_WPF + 1 of them whatever the job's size); job >= n_jobs: all out of range.
__device__ __forceinline__ void tile_wload(const NetBufs &nb, const TileSched &ts, int job, int lane, WReg &r)
{
    const bool real = job < ts.n_jobs;
    const int jj = real ? job : 0;
    const int w_off = sched_get(ts, 0, jj), nkb = real ? sched_get(ts, 1, jj) : 0, b_col = sched_get(ts, 7, jj);
    const int ncols = real ? sched_get(ts, 8, jj) : 0;
    // a descriptor of exactly this job's blocks: the hardware drops the loads behind them, so the addresses are the
    // same for every job (one register and immediate offsets) where a per-load select cost three instructions each
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(nb.wbase) + w_off, 0, nkb * 1024, 0x00020000);
#pragma unroll
    for (int i = 0; i < MZ_WPF; i++)
        r.w[i] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16 + i * 1024, 0, 0));
    const int boff = ncols > 0 ? (lane >> 4) * 16 : 0x7ffffff0;   // (the bias array is padded to whole tiles of 16)
    r.bias = as_float4(__builtin_amdgcn_raw_buffer_load_b128(nb.b, boff, b_col * 4, 0));
}

// Workgroup barrier that orders LDS traffic only: global loads in flight (the weights of a later job) stay in flight
// across it, where __syncthreads would wait for them.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// NKB blocks of one job, straight-line: the activations of block i + 1 are read while block i multiplies.
template <int NKB, int HALVES>
__device__ __forceinline__ void mfma_blocks(const float *a0, const float *a1, const float4 (&w)[MZ_WPF],
                                            const int (&kb)[MZ_WPF], f32x4 &acc0, f32x4 &acc1)
{
    float4 x0 = *reinterpret_cast<const float4 *>(a0 + kb[0] * 16);
    float4 x1 = {0, 0, 0, 0};
    if constexpr (HALVES == 2) x1 = *reinterpret_cast<const float4 *>(a1 + kb[0] * 16);
#pragma unroll
    for (int i = 0; i < NKB; i++) {
        float4 n0 = x0, n1 = x1;
        if (i + 1 < NKB) {
            n0 = *reinterpret_cast<const float4 *>(a0 + kb[i + 1] * 16);
            if constexpr (HALVES == 2) n1 = *reinterpret_cast<const float4 *>(a1 + kb[i + 1] * 16);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the reads of block i + 1 ahead of block i's matrix instructions
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].x, x0.x, acc0, 0, 0, 0);
        if constexpr (HALVES == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].x, x1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].y, x0.y, acc0, 0, 0, 0);
        if constexpr (HALVES == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].y, x1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].z, x0.z, acc0, 0, 0, 0);
        if constexpr (HALVES == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].z, x1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].w, x0.w, acc0, 0, 0, 0);
        if constexpr (HALVES == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].w, x1.w, acc1, 0, 0, 0);
        x0 = n0; x1 = n1;
    }
}

// The same with a call behind every block - after(integral_constant<int, i>) behind the matrix instructions of block i,
// after(integral_constant<int, 100 + NKB>) behind the last block - written as a recursion so that the block's number
// is a constant inside `after` (the callers ask for later weights there).
template <int NKB, int HALVES, int I = 0, class After>
__device__ __forceinline__ void mfma_blocks_hooked(const float *a0, const float *a1, const float4 (&w)[MZ_WPF],
                                                   const int (&kb)[MZ_WPF], f32x4 &acc0, f32x4 &acc1, float4 x0, float4 x1,
                                                   After &after)
{
    if constexpr (I < NKB) {
        float4 n0 = x0, n1 = x1;
        if constexpr (I + 1 < NKB) {
            n0 = *reinterpret_cast<const float4 *>(a0 + kb[I + 1] * 16);
            if constexpr (HALVES == 2) n1 = *reinterpret_cast<const float4 *>(a1 + kb[I + 1] * 16);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the reads of block I + 1 ahead of block I's matrix instructions
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[I].x, x0.x, acc0, 0, 0, 0);
        if constexpr (HALVES == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[I].x, x1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[I].y, x0.y, acc0, 0, 0, 0);
        if constexpr (HALVES == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[I].y, x1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[I].z, x0.z, acc0, 0, 0, 0);
        if constexpr (HALVES == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[I].z, x1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[I].w, x0.w, acc0, 0, 0, 0);
        if constexpr (HALVES == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[I].w, x1.w, acc1, 0, 0, 0);
        after(std::integral_constant<int, I>());
        mfma_blocks_hooked<NKB, HALVES, I + 1>(a0, a1, w, kb, acc0, acc1, n0, n1, after);
    } else {
        after(std::integral_constant<int, 100 + NKB>());
    }
}

// n blocks (0..MZ_WPF) with the calls described above (n = 0: only the closing call)
template <int HALVES, class After>
__device__ __forceinline__ void mfma_round_hooked(int n, const float *a0, const float *a1, const float4 (&w)[MZ_WPF],
                                                  const int (&kb)[MZ_WPF], f32x4 &acc0, f32x4 &acc1, After after)
{
    const float4 x0 = *reinterpret_cast<const float4 *>(a0 + kb[0] * 16);
    float4 x1 = {0, 0, 0, 0};
    if constexpr (HALVES == 2) x1 = *reinterpret_cast<const float4 *>(a1 + kb[0] * 16);
    switch (n) {
    case 1: mfma_blocks_hooked<1, HALVES>(a0, a1, w, kb, acc0, acc1, x0, x1, after); break;
    case 2: mfma_blocks_hooked<2, HALVES>(a0, a1, w, kb, acc0, acc1, x0, x1, after); break;
    case 3: mfma_blocks_hooked<3, HALVES>(a0, a1, w, kb, acc0, acc1, x0, x1, after); break;
    case 4: mfma_blocks_hooked<4, HALVES>(a0, a1, w, kb, acc0, acc1, x0, x1, after); break;
    case 5: mfma_blocks_hooked<5, HALVES>(a0, a1, w, kb, acc0, acc1, x0, x1, after); break;
    case 6: mfma_blocks_hooked<6, HALVES>(a0, a1, w, kb, acc0, acc1, x0, x1, after); break;
    case 7: mfma_blocks_hooked<7, HALVES>(a0, a1, w, kb, acc0, acc1, x0, x1, after); break;
    case 8: mfma_blocks_hooked<8, HALVES>(a0, a1, w, kb, acc0, acc1, x0, x1, after); break;
    case 9: mfma_blocks_hooked<9, HALVES>(a0, a1, w, kb, acc0, acc1, x0, x1, after); break;
    case 10: mfma_blocks_hooked<10, HALVES>(a0, a1, w, kb, acc0, acc1, x0, x1, after); break;
    default: mfma_blocks_hooked<0, HALVES>(a0, a1, w, kb, acc0, acc1, x0, x1, after); break;
    }
}

template <int HALVES>
__device__ __forceinline__ void mfma_round(int n, const float *a0, const float *a1, const float4 (&w)[MZ_WPF],
                                           const int (&kb)[MZ_WPF], f32x4 &acc0, f32x4 &acc1)
{
    switch (n) {
    case 1: mfma_blocks<1, HALVES>(a0, a1, w, kb, acc0, acc1); break;
    case 2: mfma_blocks<2, HALVES>(a0, a1, w, kb, acc0, acc1); break;
    case 3: mfma_blocks<3, HALVES>(a0, a1, w, kb, acc0, acc1); break;
    case 4: mfma_blocks<4, HALVES>(a0, a1, w, kb, acc0, acc1); break;
    case 5: mfma_blocks<5, HALVES>(a0, a1, w, kb, acc0, acc1); break;
    case 6: mfma_blocks<6, HALVES>(a0, a1, w, kb, acc0, acc1); break;
    case 7: mfma_blocks<7, HALVES>(a0, a1, w, kb, acc0, acc1); break;
    case 8: mfma_blocks<8, HALVES>(a0, a1, w, kb, acc0, acc1); break;
    case 9: mfma_blocks<9, HALVES>(a0, a1, w, kb, acc0, acc1); break;
    case 10: mfma_blocks<10, HALVES>(a0, a1, w, kb, acc0, acc1); break;
    default: break;
    }
}

// One scheduled job on the LDS-resident tile; `first`: its first MZ_WPF weight blocks and bias, loaded ahead.
// `next`: receives the loads of the wave's following job (tile_wload's, issued here two behind every block that
// multiplies and the rest behind the last: a compute unit's vector-memory path takes one 1 KB load per 16 cycles, and
// eight waves that each issue eleven of them when a level begins stand 1.4 k cycles in front of their first matrix
// instruction).
template <int TM>
__device__ __forceinline__ void tile_job(const MfmaNet &net, const NetBufs &nb, float *act, const TileSched &ts, int job,
                                         int lane, const WReg &first, WReg &next)
{
    const int w_off = sched_get(ts, 0, job), nkb = sched_get(ts, 1, job), mode = sched_get(ts, 2, job);
    const int in_off = sched_get(ts, 3, job), out_col = sched_get(ts, 4, job), skip_col = sched_get(ts, 5, job);
    const int actf = sched_get(ts, 6, job), ncols = sched_get(ts, 8, job);
    const int p[3] = {sched_get(ts, 10, job), sched_get(ts, 11, job), sched_get(ts, 12, job)};
    int kb[MZ_WPF];
#pragma unroll
    for (int i = 0; i < MZ_WPF; i++) kb[i] = (p[i / 4] >> (8 * (i % 4))) & 0xff;
#ifdef MZ_STAMPS
    // diagnostic build: cycles of a job before / in / after its matrix instructions, summed per wave of block 0
    long long jt_ = clock64();
    const bool jw_ = blockIdx.x == 0 && lane == 0;
    const int jb_ = 7200 + 8 * (int)(threadIdx.x >> 6);
#define MZ_JSTAMP(k) if (jw_) { long long n_ = clock64(); mz_dbg_stamps[jb_ + (k)] += n_ - jt_; jt_ = n_; }
#else
#define MZ_JSTAMP(k)
#endif
    const int row0 = mode == 2 ? 16 : 0;
    float *base = act + (row0 + (lane & 15)) * net.stride;
    const float *a0 = base + in_off + (lane >> 4) * 4;
    const float *a1 = a0 + 16 * net.stride;
    const bool both = TM == 32 && mode == 0;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const int n0 = nkb < MZ_WPF ? nkb : MZ_WPF;
    MZ_JSTAMP(0)
    {
        const bool nreal = job + 1 < ts.n_jobs;
        const int nj = nreal ? job + 1 : 0;
        const int nw_off = sched_get(ts, 0, nj), nnkb = nreal ? sched_get(ts, 1, nj) : 0, nb_col = sched_get(ts, 7, nj);
        const int nncols = nreal ? sched_get(ts, 8, nj) : 0;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(nb.wbase) + nw_off, 0, nnkb * 1024, 0x00020000);
        const int boff = nncols > 0 ? (lane >> 4) * 16 : 0x7ffffff0;
        auto ask = [&](auto I) {
            constexpr int v = decltype(I)::value;
            if constexpr (v < 100) {
                static_for<MZ_TILE_ASK>([&](auto J) {
                    constexpr int k = MZ_TILE_ASK * v + decltype(J)::value;
                    if constexpr (k < MZ_WPF) next.w[k] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16 + k * 1024, 0, 0));
                });
            } else {
                static_for<MZ_WPF>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    if constexpr (k >= MZ_TILE_ASK * (v - 100)) next.w[k] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16 + k * 1024, 0, 0));
                });
                next.bias = as_float4(__builtin_amdgcn_raw_buffer_load_b128(nb.b, boff, nb_col * 4, 0));
            }
        };
        if (both) mfma_round_hooked<2>(n0, a0, a1, first.w, kb, acc0, acc1, ask);
        else mfma_round_hooked<1>(n0, a0, a1, first.w, kb, acc0, acc1, ask);
    }
#ifdef MZ_STAMPS
    if (acc0[0] == 12345.678f && acc1[3] == 1.0f) jt_++;   // the stamp waits for the results
#endif
    MZ_JSTAMP(1)
    if (nkb > MZ_WPF) {  // wide layers: further rounds, loaded here (their latency is exposed)
        const int32_t *kx = net.kidx + sched_get(ts, 9, job);
        for (int k0 = MZ_WPF; k0 < nkb; k0 += MZ_WPF) {
            float4 w[MZ_WPF];
            int kb2[MZ_WPF];
#pragma unroll
            for (int i = 0; i < MZ_WPF; i++) {
                const bool has = k0 + i < nkb;
                const int voff = has ? lane * 16 + (k0 + i) * 1024 : 0x7ffffff0;
                w[i] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(nb.w, voff, w_off, 0));
                kb2[i] = has ? kx[k0 + i] : 0;
            }
            const int n = nkb - k0 < MZ_WPF ? nkb - k0 : MZ_WPF;
            if (both) mfma_round<2>(n, a0, a1, w, kb2, acc0, acc1);
            else mfma_round<1>(n, a0, a1, w, kb2, acc0, acc1);
        }
    }
    // C/D layout of 16x16x4 with the weights as first operand: tree = lane & 15, output column = (lane >> 4) * 4 + register
    const int c0 = (lane >> 4) * 4;
    if (c0 < ncols) {
        const float bb[4] = {first.bias.x, first.bias.y, first.bias.z, first.bias.w};
        float *o0 = act + (row0 + (lane & 15)) * net.stride + c0, *o1 = o0 + 16 * net.stride;
        float v0[4], v1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { v0[j] = acc0[j] + bb[j]; v1[j] = acc1[j] + bb[j]; }
        if (skip_col >= 0) {   // (the residual tensor is padded to whole tiles like every tensor of the row)
            const float4 s0 = *reinterpret_cast<const float4 *>(o0 + skip_col);
            v0[0] += s0.x; v0[1] += s0.y; v0[2] += s0.z; v0[3] += s0.w;
            if (both) {
                const float4 s1 = *reinterpret_cast<const float4 *>(o1 + skip_col);
                v1[0] += s1.x; v1[1] += s1.y; v1[2] += s1.z; v1[3] += s1.w;
            }
        }
        // the activation is the job's (wave-uniform): one branch per job, so that a ReLU layer never pays for expm1
        if (actf == 1) {
#pragma unroll
            for (int j = 0; j < 4; j++) { v0[j] = v0[j] > 0.0f ? v0[j] : 0.0f; v1[j] = v1[j] > 0.0f ? v1[j] : 0.0f; }
        } else if (actf == 2) {
#pragma unroll
            for (int j = 0; j < 4; j++) { v0[j] = apply_act(v0[j], 2); if (both) v1[j] = apply_act(v1[j], 2); }
        }
        if (c0 + 4 <= ncols) {
            *reinterpret_cast<float4 *>(o0 + out_col) = make_float4(v0[0], v0[1], v0[2], v0[3]);
            if (both) *reinterpret_cast<float4 *>(o1 + out_col) = make_float4(v1[0], v1[1], v1[2], v1[3]);
        } else {   // the tile's last, partial group of columns: the padding behind the tensor stays as it is
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (c0 + j < ncols) {
                    o0[out_col + j] = v0[j];
                    if (both) o1[out_col + j] = v1[j];
                }
            }
        }
    }
    MZ_JSTAMP(2)
#ifdef MZ_STAMPS
    if (jw_) mz_dbg_stamps[jb_ + 3] += 1;
#endif
#undef MZ_JSTAMP
}

// Shared pieces of the two kernels below.  A block of MZ_NET_NT threads owns a tile of MZ_NET_TM trees;
// tree i of the tile is also served by the 16-lane group i (threads 16 i .. 16 i + 15) for everything
// that is per tree: staging its inputs, the epilogue, and (whole-search kernel) its tree operations.
static_assert(MZ_NET_NT / MZ_NET_TM == 16, "one 16-lane group per tree of the tile");

template <int TM>
__device__ __forceinline__ void tile_zero(float *act, int stride, int tid)
{
    // Zero what the program reads but never writes: the padding behind every tensor (weights are zero
    // there, but 0 * garbage must stay finite).  Cheapest is the whole tile, as 16-byte stores.
    float4 z = {0, 0, 0, 0};
    float4 *a4 = reinterpret_cast<float4 *>(act);
    for (int i = tid; i < TM * stride / 4; i += TM * 16) a4[i] = z;
}

// inputs of one tree: hidden state of its selected PARENT and the encoded action (mcts/mcts.py:240-243).  Stored hidden
// states are [channel][cell]; a plane-major network wants [cell][channel] in its activation row.
template <int LANES = 16>
__device__ __forceinline__ void tile_stage_inputs(const MfmaNet &net, float *row, const float *ph, int action, int a)
{
    const int H = net.io[6], asz = net.io[8];
    if (net.plane_major) {   // lane = channel: no index arithmetic per element, consecutive lanes store consecutive floats
        const int C = net.io[10], P = H / C;
        constexpr int NB = 9;   // a 3 x 3 board's cells in one batch of loads (the plain loop waits for memory every four)
        for (int ch = a; ch < C; ch += LANES) {
            const float *src = ph + ch * P;
            float *dst = row + net.io[0] + ch;
            for (int c0 = 0; c0 < P; c0 += NB) {
                float v[NB];
#pragma unroll
                for (int u = 0; u < NB; u++) v[u] = src[c0 + u < P ? c0 + u : P - 1];
#pragma unroll
                for (int u = 0; u < NB; u++) if (c0 + u < P) dst[(c0 + u) * C] = v[u];
            }
        }
    } else if ((H & 3) == 0) {
        const float4 *p4 = reinterpret_cast<const float4 *>(ph);
        float4 *r4 = reinterpret_cast<float4 *>(row + net.io[0]);
        for (int j = a; j < H / 4; j += LANES) r4[j] = p4[j];
    } else {
        for (int j = a; j < H; j += LANES) row[net.io[0] + j] = ph[j];
    }
    for (int j = a; j < asz; j += LANES) row[net.io[1] + j] = j == action ? 1.0f : 0.0f;
}

// all levels of the program for a tile of TM (32 or 16) trees; ends with a workgroup barrier.  `w0`: the loads of the
// wave's first job, issued by the caller with tile_wload(.., 0, ..) before it waited for the tile's inputs.
template <int TM>
__device__ __forceinline__ void tile_run_levels(const MfmaNet &net, const NetBufs &nb, float *act, const TileSched &ts,
                                                int lane, WReg &w0)
{
    int job = 0;
#ifdef MZ_STAMPS
    long long lt_ = clock64();   // diagnostic build: cycles per level (jobs, then the wait at its barrier), wave 0 of block 0
    const bool lw_ = blockIdx.x == 0 && threadIdx.x == 0;
#endif
    // (one job ahead: requesting the weights two jobs ahead was measured slower, 4 % with a register rotation by
    // moves and 1 % with three register sets used in turn - weight latency is not what a job waits for)
    // (finishing a job's outputs behind the next job's matrix instructions, with two accumulator sets in turn, was
    // measured too: 1 % faster in k_net_step, 4 % slower in the whole-search kernels, which it pushes into spilling)
    for (int l = 0; l < net.n_levels; l++) {
        const int lend = __builtin_amdgcn_readlane(ts.lvl_end, l);
        for (; job < lend; job++) {
            WReg w1;
            tile_job<TM>(net, nb, act, ts, job, lane, w0, w1);
            w0 = w1;
        }
#ifdef MZ_STAMPS
        if (lw_ && l < 8) { long long n_ = clock64(); mz_dbg_stamps[7000 + 2 * l] += n_ - lt_; lt_ = n_; }
#endif
        lds_barrier();
#ifdef MZ_STAMPS
        if (lw_ && l < 8) { long long n_ = clock64(); mz_dbg_stamps[7000 + 2 * l + 1] += n_ - lt_; lt_ = n_; }
#endif
    }
}

// Epilogue of one tree by its group of LANES lanes: next hidden state (per-group min-max scaling,
// utils/network_utils.py:132-158) into `hidden_dst`, value and reward (support_to_scalar,
// utils/network_utils.py:75-96).  The policy logits stay in the tree's row at io[4].
// SUP = lanes that reduce the supports: the float32 sums depend on it, so every path that must agree bit for
// bit with the 16-lane groups of the tile-resident kernels uses 16 (each 16-lane row of a wider group then
// computes the same thing).
// The hidden-state half of the epilogue: the raw state at io[2] of the row, scaled per group, into `hidden_dst`.
template <int LANES>
__device__ __forceinline__ void tile_hidden_out(const MfmaNet &net, const float *row, int a, float *hidden_dst)
{
    const int H = net.io[6], ngroups = net.io[10];
    const int per = H / ngroups;
    if (ngroups >= 8) {  // many small groups (board planes): one lane per group
        const int sj = net.plane_major ? ngroups : 1, sg = net.plane_major ? 1 : per;  // element (g, j) at g * sg + j * sj
        for (int g = a; g < ngroups; g += LANES) {
            const float *raw = row + net.io[2] + g * sg;
            float mn = raw[0], mx = raw[0];
            for (int j = 1; j < per; j++) { float v = raw[j * sj]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
            float den = (mx - mn) + 1e-5f;
            for (int j = 0; j < per; j++) hidden_dst[g * per + j] = (raw[j * sj] - mn) / den;
        }
    } else {
        minmax_to<LANES>(row + net.io[2], H, ngroups, a, hidden_dst, nullptr);
    }
}

template <int LANES = 16, int SUP = LANES>
__device__ __forceinline__ void tile_epilogue(const MfmaNet &net, const float *row, int a, float *hidden_dst,
                                               double *value, double *reward)
{
    const int sup = net.io[9];
    tile_hidden_out<LANES>(net, row, a, hidden_dst);
    support_to_scalar2<SUP>(row + net.io[5], row + net.io[3], sup, a % SUP, value, reward);
}

// recurrent_inference (network.py:105-126) for every tree's pending selection, then nothing else:
// outputs go to the leaf's hidden slot and to the per-tree logits/value/reward buffers that
// k_expand_backup consumes.
// TM = trees per block: 32, or 16 for networks whose 32-tree activation tile exceeds LDS.
template <int TM>
__global__ __launch_bounds__(TM * 16) void k_net_step(Dev d, MfmaNet net, int leaf, float *out_logits,
                                                   double *out_value, double *out_reward)
{
    extern __shared__ __align__(16) float act[];  // [TM][stride]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = tid / 16, a = tid % 16, t = blockIdx.x * TM + i;
    const int H = net.io[6], A = net.io[7];
    const bool live = t < d.B;
    float *row = act + i * net.stride;
#ifdef MZ_STAMPS
    long long kst_ = clock64();
#endif
    tile_zero<TM>(act, net.stride, tid);
    __syncthreads();
    MZ_KSTAMP(0)
    const NetBufs nb = net_bufs(net);
    TileSched ts;
    tile_sched_load<TM>(net, wave, lane, ts);
    MZ_KSTAMP(1)
    if (live) {
        int parent = d.sel_parent[t], action = d.sel_action[t];
        tile_stage_inputs(net, row, d.hidden + ((long long)t * d.L.N + parent) * H, action, a);
    }
    WReg wf;
    tile_wload(nb, ts, 0, lane, wf);
    MZ_KSTAMP(2)
    lds_barrier();
    MZ_KSTAMP(3)
    tile_run_levels<TM>(net, nb, act, ts, lane, wf);
    MZ_KSTAMP(4)
    if (!live) return;
    double value, reward;
    tile_epilogue(net, row, a, d.hidden + ((long long)t * d.L.N + leaf) * H, &value, &reward);
    MZ_KSTAMP(5)
    for (int j = a; j < A; j += 16) out_logits[(long long)t * A + j] = row[net.io[4] + j];
    if (a == 0) { out_value[t] = value; out_reward[t] = reward; }
}

// ---------------------------------------------------------------------------------------------------
// Level-by-level evaluation with the activations in global memory, for batches too small to fill the chip with one tile
// of trees per block, for networks whose tile does not fit in LDS or has no schedule, and for more than 16 actions:
// the grid covers (blocks of a few column tiles of one step) x (tiles of 32 trees), so a level of a wide network is
// hundreds of blocks even for a thousand trees (k_stage_inputs, k_net_level_rows below).  Same k order per output as
// the tile-resident kernels, hence the same results.

// Epilogue of the per-level path: one group of LANES (16 or 64) lanes per tree reads its row of `acts`.
// The reductions are order-insensitive except for the float32 sums of the supports, whose order depends
// on the lanes that share them: supports of up to 64 bins are always reduced by 16 lanes (the tile-resident
// kernels' results bit for bit); longer ones by the whole wave when LANES is 64 (faster, last-bit differences).
template <int LANES>
__global__ __launch_bounds__(256) void k_net_epilogue(Dev d, MfmaNet net, int leaf, const float *acts, float *out_logits,
                                                       double *out_value, double *out_reward)
{
    const int t = (int)(blockIdx.x * (256 / LANES) + threadIdx.x / LANES), a = threadIdx.x % LANES;
    if (t >= d.B) return;
    const int H = net.io[6], A = net.io[7];
    const float *row = acts + (long long)t * net.stride;
    double value, reward;
    float *hidden_dst = d.hidden + ((long long)t * d.L.N + leaf) * H;
    if (LANES == 64 && net.io[9] <= 64) tile_epilogue<LANES, 16>(net, row, a, hidden_dst, &value, &reward);
    else tile_epilogue<LANES, LANES>(net, row, a, hidden_dst, &value, &reward);
    for (int j = a; j < A; j += LANES) out_logits[(long long)t * A + j] = row[net.io[4] + j];
    if (a == 0) { out_value[t] = value; out_reward[t] = reward; }
}

// ---------------------------------------------------------------------------------------------------
// initial_inference (network.py:86-103) on the level kernels: the representation program turns observations into raw
// hidden states, which are scaled per group (utils/network_utils.py:132-158) and handed to the prediction program;
// policy logits and the value (support_to_scalar, utils/network_utils.py:75-96) come out of its row.  One wave per
// root; the matrix work in between is k_net_level_rows on each program's own activation rows.

// observations [B][io[11]] into the rows of the representation program (its input tensor starts at io[0])
static __global__ __launch_bounds__(256) void k_root_stage_obs(int B, MfmaNet net, const float *obs, float *acts)
{
    const int t = (int)(blockIdx.x * 4 + threadIdx.x / 64), a = threadIdx.x % 64;
    if (t >= B) return;
    const int n = net.io[11];
    float *row = acts + (long long)t * net.stride + net.io[0];
    for (int j = a; j < n; j += 64) row[j] = obs[(long long)t * n + j];
}

// the scaled hidden state of every root: out of the representation program's row into out_hidden [B][H]
// (LANES per root: 16 when the state is scaled per board plane - one lane per plane -, else a whole wave)
template <int LANES>
__global__ __launch_bounds__(256) void k_root_hidden(int B, MfmaNet rep, const float *acts, float *out_hidden)
{
    const int t = (int)(blockIdx.x * (256 / LANES) + threadIdx.x / LANES), a = threadIdx.x % LANES;
    if (t >= B) return;
    tile_hidden_out<LANES>(rep, acts + (long long)t * rep.stride, a, out_hidden + (long long)t * rep.io[6]);
}

// ... and into the rows of the prediction program (io[8] of that program is 0: no action beside the state)
static __global__ __launch_bounds__(256) void k_root_stage_hidden(int B, MfmaNet pred, const float *hidden, float *acts)
{
    const int t = (int)(blockIdx.x * 4 + threadIdx.x / 64), a = threadIdx.x % 64;
    if (t >= B) return;
    tile_stage_inputs<64>(pred, acts + (long long)t * pred.stride, hidden + (long long)t * pred.io[6], 0, a);
}

// policy logits and value of every root out of the prediction program's row
static __global__ __launch_bounds__(256) void k_root_outputs(int B, MfmaNet pred, const float *acts, float *out_logits, double *out_value)
{
    const int t = (int)(blockIdx.x * 4 + threadIdx.x / 64), a = threadIdx.x % 64;
    if (t >= B) return;
    const int A = pred.io[7];
    const float *row = acts + (long long)t * pred.stride;
    for (int j = a; j < A; j += 64) out_logits[(long long)t * A + j] = row[pred.io[4] + j];
    double value, twin;
    support_to_scalar2<64>(row + pred.io[5], row + pred.io[5], pred.io[9], a, &value, &twin);
    if (a == 0) out_value[t] = value;
}

// ---------------------------------------------------------------------------------------------------
// Whole searches level by level (networks too wide for a tile of trees in LDS, or more than 16 actions): per
// simulation one launch per network level plus one tree launch, for a RANGE of trees [row_first, row_first + 32 *
// gridDim.x) so that several ranges can run on streams of their own, one range's matrix work beside another's tree work.

// One level for a range of trees, inputs already in `acts` (the tree kernel below stages them).  A thread asks for its
// share of the trees' input columns, then its wave for the weights of its column tile (up to MZ_LVL_KB blocks, a fixed
// number of buffer loads: blocks it does not have are addressed out of range and cost no memory access; wider layers
// fetch the rest in rounds later); the inputs go through LDS, and the weight blocks arrive while they do and while the
// first blocks multiply.  Same k order per output as tile_job.
template <int KB>
__global__ __launch_bounds__(MZ_LVL_WAVES * 64) void k_net_level_rows(MfmaNet net, const int32_t *desc, float *acts,
                                                                      int row_first, int level)
{
    static_assert(KB == MZ_WPF || KB == 2 * MZ_WPF, "one or two rounds of weight blocks");
#ifdef MZ_STAMPS
    // diagnostic build: bits 8.. of `level` ask the kernel to stop early (1: at once, 2: after the barrier, 3: before
    // the output stores), which prices the launch itself and each stage
    const int stop_ = level >> 8;
    level &= 0xff;
    if (stop_ == 1) return;
    long long kst_ = clock64();
    const int lin_ = blockIdx.x * gridDim.y + blockIdx.y;
    const int slot_ = (threadIdx.x == 0 && lin_ < 256 && level < 2) ? level * 256 + lin_ : -1;
#endif
    constexpr int PT = MZ_LVL_WAVES * 64 / 32;  // threads that copy one tree's inputs
    extern __shared__ __align__(16) float tile[];  // [32][ts]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // everything the block needs to know in one descriptor (MZ_LVL_DESC ints, see mz_set_network_mfma): one round
    // trip to memory between the launch and the first weight load (passing the descriptors inside the kernel
    // arguments instead was measured: 0.7 % on the 1024-tree search, not kept)
    // grid = (tiles of 32 trees, blocks of column tiles): workgroups go to the eight XCDs in turn by their linear index,
    // the tree tile is the fast one, so a tile's rows of `acts` are read and written through one XCD's L2 by all the
    // column blocks that use them (the weights, needed by every tile, travel to every L2 either way)
    const int job_ = blockIdx.y, tile_ = blockIdx.x;
    // (whatever depends on kernel arguments alone is computed here, in front of the descriptor: the compiler loads an
    // argument in the block of its first use, and an argument first used behind the descriptor costs one more round trip
    // to memory in front of the first load of data)
    const NetBufs nb = net_bufs(net);
    const int t0 = row_first + tile_ * 32;
    const float *in_row = acts + (long long)(t0 + tid / PT) * net.stride;
    // (... and the arguments are tied into the descriptor's address - by a term that is zero for every valid launch
    // (positive stride and first row, addresses below 2^63) but not for the compiler - because computing with them here
    // is not enough: the compiler moves such computations, and the argument loads with them, down to their uses)
    const int tie_ = (net.stride >> 31) | (row_first >> 31) | (int)(reinterpret_cast<uintptr_t>(net.wfrag) >> 63) |
                     (int)(reinterpret_cast<uintptr_t>(net.bias) >> 63) | (int)(reinterpret_cast<uintptr_t>(acts) >> 63);
#ifdef MZ_STAMPS
    // diagnostic build: when the kernel arguments have arrived (slot 4) and when the descriptor has (slot 5)
    if (slot_ >= 0 && tie_ == 0) mz_dbg_stamps[slot_ * 8 + 4] = clock64() - kst_;
#endif
    const int32_t *q = desc + MZ_LVL_DESC * job_ + tie_;
    // (every field is read here, before the kernel's first store: behind a store through a generic pointer the compiler
    // no longer takes these wave-uniform reads for scalar loads, and a vector load's wait covers the weights in flight)
    const int KK = q[0], ts = KK + 4, in_off = q[1];
    const int q_n = q[2], q_out = q[3], q_skip = q[4], q_act = q[5], q_bias = q[6], q_ct0 = q[8], q_dense = q[9];
    // (the wave's own three fields at an address that does not depend on the block's: one round trip to memory for both;
    // the entries of waves a block does not have are zero)
    const int32_t *qw = q + 10 + 3 * wave;
    const int w_off = qw[0], nkb_w = qw[1], kx_off = qw[2];
    const bool active = wave < q[7];
    const int nkb = active ? nkb_w : 0;
#ifdef MZ_STAMPS
    if (slot_ >= 0 && nkb + KK + kx_off >= 0) mz_dbg_stamps[slot_ * 8 + 5] = clock64() - kst_;
#endif
    int ka[MZ_WPF], kb[MZ_WPF];
    if (q_dense) {   // every column tile of the block multiplies input blocks 0, 1, 2, ...: no index list to fetch
#pragma unroll
        for (int i = 0; i < MZ_WPF; i++) { ka[i] = i; kb[i] = MZ_WPF + i; }
    } else {
        const int32_t *kx = net.kidx + kx_off;
#pragma unroll
        for (int i = 0; i < MZ_WPF; i++) {
            ka[i] = i < nkb ? kx[i] : 0;
            kb[i] = (KB > MZ_WPF && MZ_WPF + i < nkb) ? kx[MZ_WPF + i] : 0;
        }
    }
    // The trees' inputs are asked for FIRST and the weights behind them: loads come back in the order they were issued,
    // and every wave needs the whole input tile (through LDS, behind the barrier) before its first matrix instruction,
    // but its weight blocks only one after the other - so the weights stream in while the inputs are parked in LDS and
    // while the first blocks multiply.  (Issued the other way round, the copy waited for all the weights.)  A thread's
    // loads leave together, indices past the end of the row clamped: NB_IN of them cover layers of up to 384 inputs.
    constexpr int NB_IN = (96 + PT - 1) / PT;
    const int part_ = tid % PT, n4 = KK / 4;
    const f32x4 *src4 = reinterpret_cast<const f32x4 *>(in_row + in_off);
    f32x4 *d4 = reinterpret_cast<f32x4 *>(tile + (tid / PT) * ts);
    // (wider layers: the rest of a row goes first, round by round - with loads of unknown number in flight the compiler
    // would make the first matrix instruction wait for every load, the weights included)
    for (int j0 = part_ + NB_IN * PT; j0 < n4; j0 += NB_IN * PT) {
        f32x4 v[NB_IN];
        static_for<NB_IN>([&](auto U) {
            constexpr int u = decltype(U)::value;
            v[u] = src4[j0 + u * PT < n4 ? j0 + u * PT : n4 - 1];
        });
        __builtin_amdgcn_sched_barrier(0);
        static_for<NB_IN>([&](auto U) {
            constexpr int u = decltype(U)::value;
            if (j0 + u * PT < n4) d4[j0 + u * PT] = v[u];
        });
    }
    f32x4 vin[NB_IN];   // (a native vector type: arrays of it become registers; rows past the batch exist in `acts`, zero)
    static_for<NB_IN>([&](auto U) {
        constexpr int u = decltype(U)::value;
        vin[u] = src4[part_ + u * PT < n4 ? part_ + u * PT : n4 - 1];
    });
    __builtin_amdgcn_sched_barrier(0);   // (the compiler would move these loads down to their stores, behind the weights)
    float4 wa[MZ_WPF], wb[MZ_WPF];
    // (a descriptor of exactly this wave's blocks: the hardware drops the loads behind them)
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(nb.wbase) + w_off, 0, nkb * 1024, 0x00020000);
    // Only the first MZ_LVL_AHEAD blocks are asked for here; block i + MZ_LVL_AHEAD is asked for behind the matrix
    // instructions of block i.  A compute unit's vector-memory path takes 64 bytes per cycle: a block's 26 loads of 1 KB
    // per wave occupy it for 3.3 k cycles, during which - issued all at once - its waves sat in front of the barrier.
    constexpr int AHEAD = MZ_LVL_AHEAD;
    static_assert(AHEAD >= 1 && AHEAD <= MZ_WPF, "blocks asked for before the barrier");
    auto wload = [&](int k) { return as_float4(__builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16 + k * 1024, 0, 0)); };
#pragma unroll
    for (int i = 0; i < AHEAD; i++) wa[i] = wload(i);
    // (result layout as in tile_job: tree = lane & 15, output columns (lane >> 4) * 4 .. + 3 down the registers)
    const int col = (q_ct0 + (active ? wave : 0)) * 16 + (lane >> 4) * 4, N = q_n;
    float4 bias4 = {0, 0, 0, 0};
    if (active && col < N) bias4 = *reinterpret_cast<const float4 *>(net.bias + q_bias + col);   // padded to whole tiles
    MZ_SSTAMP(slot_, 0)
    {   // the inputs' way into LDS (the weights stay in flight)
        static_for<NB_IN>([&](auto U) {
            constexpr int u = decltype(U)::value;
            if (part_ + u * PT < n4) d4[part_ + u * PT] = vin[u];
        });
    }
    MZ_SSTAMP(slot_, 1)
    __syncthreads();
    MZ_SSTAMP(slot_, 2)
#ifdef MZ_STAMPS
    if (stop_ == 2) { if (wa[0].x == 12345.678f && bias4.x == 1.0f) acts[0] = 0.0f; return; }
#endif
    if (!active) return;
    const float *a0 = tile + (lane & 15) * ts + (lane >> 4) * 4, *a1 = a0 + 16 * ts;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    // (blocks the wave does not have are addressed out of range: dropped by the hardware, zeros in the registers)
    mfma_round_hooked<2>(nkb < MZ_WPF ? nkb : MZ_WPF, a0, a1, wa, ka, acc0, acc1, [&](auto I) {
        constexpr int k = decltype(I)::value + AHEAD;
        if constexpr (k < MZ_WPF) wa[k] = wload(k);
        else if constexpr (KB > MZ_WPF && k < 100) wb[k - MZ_WPF] = wload(k);
    });
    if constexpr (KB > MZ_WPF) {
        if (nkb > MZ_WPF)
            mfma_round_hooked<2>((nkb < KB ? nkb : KB) - MZ_WPF, a0, a1, wb, kb, acc0, acc1, [&](auto I) {
                constexpr int k = decltype(I)::value + AHEAD;
                if constexpr (k < MZ_WPF) wb[k] = wload(MZ_WPF + k);
            });
        // layers wider than KB blocks: further rounds, requested here (their latency is exposed)
        for (int k0 = KB; k0 < nkb; k0 += MZ_WPF) {
#pragma unroll
            for (int i = 0; i < MZ_WPF; i++) {
                const int k = k0 + i;
                wa[i] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16 + k * 1024, 0, 0));
                ka[i] = k < nkb ? (q_dense ? k : net.kidx[kx_off + k]) : 0;
            }
            mfma_round<2>(nkb - k0 < MZ_WPF ? nkb - k0 : MZ_WPF, a0, a1, wa, ka, acc0, acc1);
        }
    }
#ifdef MZ_STAMPS
    if (acc0[0] == 12345.678f) kst_ += 1;   // the stamp below waits for the matrix instructions
#endif
    MZ_SSTAMP(slot_, 3)
#ifdef MZ_STAMPS
    if (stop_ == 3) { if (acc0[1] == 12345.678f) acts[0] = 0.0f; return; }
#endif
    if (col < N) {
        const int out_off = q_out, skip_off = q_skip, actf = q_act;
        const float bb[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
        float *o0 = acts + (long long)(t0 + (lane & 15)) * net.stride + col, *o1 = o0 + 16 * (long long)net.stride;
        float v0[4], v1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { v0[j] = acc0[j] + bb[j]; v1[j] = acc1[j] + bb[j]; }
        if (skip_off >= 0) {
            const float4 s0 = *reinterpret_cast<const float4 *>(o0 + skip_off), s1 = *reinterpret_cast<const float4 *>(o1 + skip_off);
            v0[0] += s0.x; v0[1] += s0.y; v0[2] += s0.z; v0[3] += s0.w;
            v1[0] += s1.x; v1[1] += s1.y; v1[2] += s1.z; v1[3] += s1.w;
        }
        // the activation is the block's: one uniform branch, so that linear and ReLU layers never pay for expm1
        if (actf == 1) {
#pragma unroll
            for (int j = 0; j < 4; j++) { v0[j] = v0[j] > 0.0f ? v0[j] : 0.0f; v1[j] = v1[j] > 0.0f ? v1[j] : 0.0f; }
        } else if (actf == 2) {
#pragma unroll
            for (int j = 0; j < 4; j++) { v0[j] = apply_act(v0[j], 2); v1[j] = apply_act(v1[j], 2); }
        }
        if (col + 4 <= N) {
            *reinterpret_cast<float4 *>(o0 + out_off) = make_float4(v0[0], v0[1], v0[2], v0[3]);
            *reinterpret_cast<float4 *>(o1 + out_off) = make_float4(v1[0], v1[1], v1[2], v1[3]);
        } else {   // the layer's last, partial group of columns
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (col + j < N) { o0[out_off + j] = v0[j]; o1[out_off + j] = v1[j]; }
        }
    }
}

// The network inputs of every tree's pending selection into its row of `acts` (once per search, after the first
// selection; later selections are staged by k_leaf_backup_select).
static __global__ __launch_bounds__(256) void k_stage_inputs(Dev d, MfmaNet net, float *acts)
{
    const int t = (int)(blockIdx.x * 4 + threadIdx.x / 64), a = threadIdx.x % 64;
    if (t >= d.B) return;
    const int parent = d.sel_parent[t], action = d.sel_action[t];
    tile_stage_inputs<64>(net, acts + (long long)t * net.stride, d.hidden + ((long long)t * d.L.N + parent) * net.io[6], action, a);
}

// tile_epilogue<64, 64> for one ungrouped hidden state of at most 64 * NH floats and supports of 65 .. 64 * NE bins, with
// the row in global memory: every value the epilogue needs is asked for up front (one trip to memory instead of one
// per reduction pass) and each exponential is taken once.  Same operations in the same order per lane as
// support_to_scalar2<64> / minmax_to<64>, hence the same bits.
template <int NE, int NH>
struct LeafRegs { float vA[NE], vB[NE], h[NH]; };

// ... the loads (issued by the caller as early as it knows the row)
template <int NE, int NH>
__device__ __forceinline__ void leaf_epilogue_load(const MfmaNet &net, const float *row, int a, LeafRegs<NE, NH> &r)
{
    const int H = net.io[6], n = net.io[9];
    const float *lgA = row + net.io[5], *lgB = row + net.io[3], *raw = row + net.io[2];
#pragma unroll
    for (int k = 0; k < NE; k++) { const int i = a + 64 * k, ii = i < n ? i : 0; r.vA[k] = lgA[ii]; r.vB[k] = lgB[ii]; }
#pragma unroll
    for (int k = 0; k < NH; k++) { const int i = a + 64 * k; r.h[k] = raw[i < H ? i : 0]; }
}

// ... and the arithmetic
template <int NE, int NH>
__device__ __forceinline__ void leaf_epilogue_regs(const MfmaNet &net, LeafRegs<NE, NH> &r, int a, float *hidden_dst,
                                                    double *value, double *reward)
{
    const int H = net.io[6], n = net.io[9], L = (n - 1) / 2;
    float (&vA)[NE] = r.vA, (&vB)[NE] = r.vB, (&h)[NH] = r.h;
    const float ninf = -__builtin_huge_valf();
    float mA = ninf, mB = ninf;
#pragma unroll
    for (int k = 0; k < NE; k++)
        if (a + 64 * k < n) { mA = vA[k] > mA ? vA[k] : mA; mB = vB[k] > mB ? vB[k] : mB; }
    auto fmaxl = [](float x, float y) { return y > x ? y : x; };
    auto fminl = [](float x, float y) { return y < x ? y : x; };
    auto faddl = [](float x, float y) { return x + y; };
    auto daddl = [](double x, double y) { return x + y; };
    mA = wave_reduce(mA, ninf, fmaxl); mB = wave_reduce(mB, ninf, fmaxl);
    float sA = 0.0f, sB = 0.0f;
#pragma unroll
    for (int k = 0; k < NE; k++)
        if (a + 64 * k < n) { vA[k] = expf(vA[k] - mA); vB[k] = expf(vB[k] - mB); sA += vA[k]; sB += vB[k]; }
    sA = wave_reduce(sA, 0.0f, faddl); sB = wave_reduce(sB, 0.0f, faddl);
    double xA = 0.0, xB = 0.0;
#pragma unroll
    for (int k = 0; k < NE; k++) {
        const int i = a + 64 * k;
        if (i < n) { xA += (double)(vA[k] / sA) * (double)(i - L); xB += (double)(vB[k] / sB) * (double)(i - L); }
    }
    xA = wave_reduce(xA, 0.0, daddl); xB = wave_reduce(xB, 0.0, daddl);
    const double y = inv_value_transform(a == 1 ? xB : xA);
    *value = wave_lane(y, 0);
    *reward = wave_lane(y, 1);
    float mn = __builtin_huge_valf(), mx = ninf;
#pragma unroll
    for (int k = 0; k < NH; k++)
        if (a + 64 * k < H) { mn = h[k] < mn ? h[k] : mn; mx = h[k] > mx ? h[k] : mx; }
    mn = wave_reduce(mn, __builtin_huge_valf(), fminl); mx = wave_reduce(mx, ninf, fmaxl);
    const float den = (mx - mn) + 1e-5f;
#pragma unroll
    for (int k = 0; k < NH; k++)
        if (a + 64 * k < H) hidden_dst[a + 64 * k] = (h[k] - mn) / den;
}

// Tree side of one simulation for trees [t_first, t_end), one wave per tree: the epilogue of the leaf evaluation that
// the level kernels left in `acts` (hidden-state scaling, supports to scalars), expansion + backup, and - unless this
// was the last simulation (select_next bit 0) - the next simulation's selection, whose network inputs go straight into
// the tree's row of `acts` for the next level launch.  select_next bit 1 (PUCT): descend over cached decisions
// (tree_ops.h puct_decide); the table must have been cleared when the search began.
template <int RL, bool GUMBEL>
__global__ __launch_bounds__(MZ_BLOCK) void k_leaf_backup_select(Dev d, MfmaNet net, int t_first, int t_end, int leaf,
                                                                  float *acts, const double *noise, int noise_max_calls,
                                                                  int select_next)
{
    __shared__ __align__(16) double s_bk[MZ_BLOCK];
    __shared__ __align__(16) double s_ct[GUMBEL ? 2 : MZ_CT_LDS];
    __shared__ __align__(16) double s_sq[GUMBEL ? 2 : MZ_SQ_LDS];
    __shared__ int32_t s_dec[GUMBEL ? 1 : MZ_BLOCK / 64][GUMBEL ? 1 : MZ_DEC_LDS];   // the waves' decision tables
    // Workgroups go to the eight XCDs in turn by their index, the level kernels keep a 32-tree tile's activation rows in
    // the L2 of XCD (tile mod 8): the eight blocks of four trees that make up a tile are given indices that land on that
    // XCD (within whole groups of 64 blocks, the two 3-bit fields of the index change places).
    unsigned vb = blockIdx.x;
    if ((vb | 63u) < gridDim.x) vb = (vb & ~63u) | ((vb & 7u) << 3) | ((vb >> 3) & 7u);
    // (kernel arguments of every 64-byte line of the argument block tied into the tree index - by a term that is zero
    // for every valid launch but not for the compiler - so that all of them are asked for at once: the compiler loads an
    // argument in front of its first use, and the lines first used later each cost a round trip to memory of their own)
    auto top_bit = [](const void *p) { return (int)(reinterpret_cast<uintptr_t>(p) >> 63); };
    const int tie_ = top_bit(d.sel_parent) | top_bit(d.path) | top_bit(d.dec) | top_bit(d.c_table) | top_bit(d.sqrt_table) |
                     top_bit(d.err) | top_bit(d.visit_total) | top_bit(d.noise_calls) | top_bit(noise) | top_bit(acts) |
                     top_bit(net.wfrag) | (d.B >> 31) | (d.L.S >> 31) | (select_next >> 31) | (leaf >> 31) |
                     (noise_max_calls >> 31) | (net.stride >> 31) | (net.io[9] >> 31);
    const int t = t_first + tie_ + (int)((vb * (unsigned)MZ_BLOCK + threadIdx.x) / 64), a = threadIdx.x % 64;
    const bool on = a < d.L.A;
    // the pending selection is asked for before the tables are staged: both waits overlap
    int parent = 0, action = 0, ntp = 0, depth = 0;
    const int H = net.io[6];
    const bool live = t < t_end;
    float *row = acts + (long long)(live ? t : t_first) * net.stride;
    // long supports, one ungrouped hidden state: the epilogue's values are asked for here as well
    const bool wide = net.io[9] > 64 && net.io[9] <= 640 && net.io[10] == 1 && H > 16 && H <= 256 && !net.plane_major;
    LeafRegs<10, 4> lr;
    float lg = 0.0f;
    // cached PUCT decisions: the tree's table travels to the wave while the epilogue and the backup run (a walk that
    // reads it from memory level by level waits for HBM at every level: the level kernels in between flush the L2)
    bool cached = false, dec_in_lds = false;
    int dec_pre[MZ_DEC_LDS / 64];
    if constexpr (!GUMBEL) {
        cached = (select_next & 3) == 3 && d.dec != nullptr;
        dec_in_lds = cached && d.L.N <= MZ_DEC_LDS;
    }
    if (live) {
        parent = d.sel_parent[t]; action = d.sel_action[t]; ntp = d.sel_ntp[t]; depth = d.sel_depth[t];
        lg = on ? row[net.io[4] + a] : 0.0f;
        if (wide) leaf_epilogue_load<10, 4>(net, row, a, lr);
        if constexpr (!GUMBEL) {
            if (dec_in_lds) {
                const int32_t *dec = d.dec + (long long)t * d.L.N;
#pragma unroll
                for (int k = 0; k < MZ_DEC_LDS / 64; k++) dec_pre[k] = a + 64 * k < d.L.N ? dec[a + 64 * k] : 0;
            }
        }
    }

#ifdef MZ_STAMPS
    long long kst_ = clock64();
    const int slot_ = (threadIdx.x == 0 && blockIdx.x < 256) ? 512 + blockIdx.x : -1;
#endif
    // The selection's tables (exploration constants, square-root factors) are not needed before the descent.  Where
    // they are small - one entry per thread - and the long sqrt table is not in use, their values are only asked for here
    // and go to LDS behind the backup, with one barrier: staged up front, their barriers held the epilogue back until
    // the slowest wave's loads had arrived.  `late`: that form; every wave then stays until that barrier.
    const double *ct = nullptr, *sqt = nullptr;
    bool sq_factors = false, late = false;
    double ct_pre = 0.0;
    if constexpr (!GUMBEL) {
        const int n_sq = (d.L.S + 2) * (d.L.S + 1);
        late = (select_next & 1) && d.L.S + 2 <= MZ_BLOCK && d.L.S + 2 <= MZ_CT_LDS && (!d.sqrt_table || n_sq > MZ_SQ_LDS);
        if (late) {
            if ((int)threadIdx.x < d.L.S + 2) ct_pre = d.c_table[threadIdx.x];
        } else {
            ct = stage_c_table<GUMBEL>(d, s_ct);
            sqt = stage_sqrt_table<GUMBEL>(d, s_sq);
        }
        // searches too long for the sqrt(n / (1 + c)) table: its two factors, sqrt(n) and 1 / sqrt(1 + c), for the cached
        // decisions (their product is within two units in the last place of the quotient's root)
        sq_factors = (select_next & 3) == 3 && d.dec != nullptr && sqt == nullptr && 2 * (d.L.S + 2) <= MZ_SQ_LDS;
        if (sq_factors && !late) {
            for (int i = threadIdx.x; i < d.L.S + 2; i += MZ_BLOCK) {
                s_sq[i] = sqrt((double)i);
                s_sq[d.L.S + 2 + i] = 1.0 / sqrt((double)(1 + i));
            }
            __syncthreads();
        }
    }
    auto stage_late = [&]() {   // every thread of the block, once
        if ((int)threadIdx.x < d.L.S + 2) {
            s_ct[threadIdx.x] = ct_pre;
            if (sq_factors) {
                s_sq[threadIdx.x] = sqrt((double)threadIdx.x);
                s_sq[d.L.S + 2 + threadIdx.x] = 1.0 / sqrt((double)(1 + threadIdx.x));
            }
        }
        __syncthreads();
        ct = s_ct;
    };
    if (t >= t_end) {
        if (late) stage_late();
        return;
    }
    MZ_SSTAMP(slot_, 0)
    if (depth < 1 || depth > d.L.S || parent < 0 || parent >= leaf) {
        if (a == 0) atomicExch(d.err, 1);
        if (late) stage_late();
        return;
    }
    const TreeMem m = tree_mem(d, t);
    float *hid = d.hidden + (long long)t * d.L.N * H;
    double value, reward;
    if (wide) leaf_epilogue_regs<10, 4>(net, lr, a, hid + (long long)leaf * H, &value, &reward);
    else if (net.io[9] <= 64) tile_epilogue<64, 16>(net, row, a, hid + (long long)leaf * H, &value, &reward);
    else tile_epilogue<64, 64>(net, row, a, hid + (long long)leaf * H, &value, &reward);
#ifdef MZ_STAMPS
    if (value == 12345.678 && lg == 3.0f) kst_ += 1;   // the stamp waits for the epilogue's results
#endif
    MZ_SSTAMP(slot_, 1)
    expand_backup<64, GUMBEL>(d, m, a, on, parent, action, ntp, depth, leaf, lg, value, reward, s_bk + (threadIdx.x / 64) * 64);
    if (a == 0) d.visit_total[t] += depth + 1;
    if (!(select_next & 1)) {
        if (a == 0) d.sel_depth[t] = -1;
        return;
    }
    __threadfence_block();
    if (late) stage_late();
    MZ_SSTAMP(slot_, 2)
    const double *nz = noise ? noise + (long long)t * noise_max_calls * d.L.A : nullptr;
    DescendCtx dc = descend_ctx<GUMBEL>(d, t, a, on, ct, sqt);
    if (sq_factors) { dc.sqrt_n = s_sq; dc.rsqrt_1c = s_sq + d.L.S + 2; }
    uint32_t calls = GUMBEL ? 0u : d.noise_calls[t];
    if (cached) {
        if constexpr (!GUMBEL) {
            // the records this simulation changed - the path and the new leaf - get their decisions again, one lane per
            // node or per group of lanes; everything else in the table still stands
            int32_t *dec = d.dec + (long long)t * d.L.N;
            int32_t *dec_lds = dec_in_lds ? s_dec[threadIdx.x / 64] : nullptr;
            if (dec_in_lds) {
#pragma unroll
                for (int k = 0; k < MZ_DEC_LDS / 64; k++) dec_lds[a + 64 * k] = dec_pre[k];
                lds_fence();
            }
            const int n_path = depth < d.L.S ? depth : d.L.S;
            auto node_of = [&](int j) { return j < n_path ? m.path[j] >> 6 : leaf; };
            const int n = n_path + 1;
            // lanes per node: as many as cover the list in one pass, but no fewer than keep a lane's share of the
            // actions in registers (then a pass waits for memory once; longer lists take several passes)
            const int A = d.L.A;
            const int g_regs = A <= MZ_PUCT_NA ? 1 : A <= 2 * MZ_PUCT_NA ? 2 : A <= 4 * MZ_PUCT_NA ? 4 : 8;
            const int g_list = n <= 4 ? 16 : n <= 8 ? 8 : n <= 16 ? 4 : n <= 32 ? 2 : 1;
            const int G = g_list > g_regs ? g_list : g_regs;
            if (G == 16) puct_decide_list<16>(d, dc, m, dec, dec_lds, n, a, node_of);
            else if (G == 8) puct_decide_list<8>(d, dc, m, dec, dec_lds, n, a, node_of);
            else if (G == 4) puct_decide_list<4>(d, dc, m, dec, dec_lds, n, a, node_of);
            else if (G == 2) puct_decide_list<2>(d, dc, m, dec, dec_lds, n, a, node_of);
            else puct_decide_list<1>(d, dc, m, dec, dec_lds, n, a, node_of);
            // the walk below reads the LDS copy; the table in memory is for the next launch
            if (dec_in_lds) lds_fence();
            else __threadfence_block();
            int misses_ = 0;
#ifdef MZ_STAMPS
            const long long t_dec_ = clock64();
#endif
            if (dec_in_lds)
                parent = descend_cached_puct<RL, true>(d, dc, m, dec, dec_lds, a, on, nz, noise_max_calls, calls, &action, &ntp,
                                                       &depth, &misses_);
            else
                parent = descend_cached_puct<RL, false>(d, dc, m, dec, dec_lds, a, on, nz, noise_max_calls, calls, &action, &ntp,
                                                        &depth, &misses_);
#ifdef MZ_STAMPS
            if (slot_ >= 0) { mz_dbg_stamps[slot_ * 8 + 6] = misses_; mz_dbg_stamps[slot_ * 8 + 7] = clock64() - t_dec_; }
#endif
        }
    } else {
        parent = descend<64, RL, GUMBEL>(d, dc, m, a, on, nz, noise_max_calls, calls, &action, &ntp, &depth);
    }
    MZ_SSTAMP(slot_, 3)
    if (a == 0) {
        if constexpr (!GUMBEL) store_noise_calls(d, t, calls, noise_max_calls);
        d.sel_parent[t] = parent; d.sel_action[t] = action; d.sel_ntp[t] = ntp; d.sel_depth[t] = depth;
    }
    tile_stage_inputs<64>(net, row, hid + (long long)parent * H, action, a);
#ifdef MZ_STAMPS
    __threadfence();
#endif
    MZ_SSTAMP(slot_, 4)
#ifdef MZ_STAMPS
    if (slot_ >= 0) mz_dbg_stamps[slot_ * 8 + 5] = depth;
#endif
}

// The whole simulation loop in one launch with leaves evaluated on the matrix cores: per simulation
// every group descends its tree and stages the leaf's inputs, the block runs the network for its
// 32-tree tile, then every group finishes its tree (epilogue, expansion, backup).  Trees stay in
// global memory; the tile's activations stay in LDS for the whole search.  RL <= 16.  TM = trees per
// block: 16 lets two blocks share a CU, so one block's matrix work overlaps the other's tree work.
template <int TM, int RL, bool GUMBEL>
__global__ __launch_bounds__(TM * 16) void k_search_mfma(Dev d, MfmaNet net, int first_leaf, int n_sims,
                                                            const double *noise, int noise_max_calls)
{
    extern __shared__ __align__(16) float act[];  // [TM][stride]
    __shared__ __align__(16) double s_bk[TM * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = tid / 16, a = tid % 16, t = blockIdx.x * TM + i;
    const int H = net.io[6], A = d.L.A;
    const bool live = t < d.B, on = a < A;
    float *row = act + i * net.stride;
    double *bk = s_bk + i * 16;
    tile_zero<TM>(act, net.stride, tid);
    const int tt = live ? t : 0;  // groups past the end of the batch only keep the barriers company
    const TreeMem m = tree_mem(d, tt);
    float *hid = d.hidden + (long long)tt * d.L.N * H;
    const double *nz = noise ? noise + (long long)tt * noise_max_calls * A : nullptr;
    DescendCtx dc = descend_ctx<GUMBEL>(d, tt, a, on, nullptr, nullptr);
    uint32_t calls = GUMBEL ? 0u : d.noise_calls[tt];
    long long visited = 0;
    const NetBufs nb = net_bufs(net);
    TileSched ts;
    tile_sched_load<TM>(net, wave, lane, ts);
    __syncthreads();
#ifdef MZ_STAMPS
    long long st_a = 0, st_b = 0, st_c = 0, st_d = 0, st_t = clock64();
#define MZ_MSTAMP(acc) { long long n_ = clock64(); acc += n_ - st_t; st_t = n_; }
#else
#define MZ_MSTAMP(acc)
#endif
    for (int s = 0; s < n_sims; s++) {
        const int leaf = first_leaf + s;
        int parent = 0, action = 0, ntp = 0, depth = 0;
        WReg wf;
        tile_wload(nb, ts, 0, lane, wf);   // (the first job's weights depend on nothing a simulation does)
        if (live) {
            parent = descend<16, RL, GUMBEL>(d, dc, m, a, on, nz, noise_max_calls, calls, &action, &ntp, &depth);
            visited += depth + 1;
            tile_stage_inputs(net, row, hid + (long long)parent * H, action, a);
        }
        MZ_MSTAMP(st_a)
        __syncthreads();
        MZ_MSTAMP(st_b)
        tile_run_levels<TM>(net, nb, act, ts, lane, wf);
        MZ_MSTAMP(st_c)
        if (live) {
            double value, reward;
            tile_epilogue(net, row, a, hid + (long long)leaf * H, &value, &reward);
            float lg = on ? row[net.io[4] + a] : 0.0f;
            expand_backup<16, GUMBEL>(d, m, a, on, parent, action, ntp, depth, leaf, lg, value, reward, bk);
            __threadfence_block();
        }
        MZ_MSTAMP(st_d)
    }
#ifdef MZ_STAMPS
    if (live && a == 0) {  // diagnostic build only: [descend+stage, barrier wait, network, epilogue+backup] cycles
        long long *o = reinterpret_cast<long long *>(d.path + (long long)t * d.L.S);
        o[0] = st_a; o[1] = st_b; o[2] = st_c; o[3] = st_d;
    }
#endif
    if (live && a == 0) {
        if constexpr (!GUMBEL) store_noise_calls(d, t, calls, noise_max_calls);
        d.visit_total[t] += visited;
    }
}

}  // namespace mz
