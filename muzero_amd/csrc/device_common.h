// Device-side building blocks of the tree kernels (gfx950, wave64).
//
// A tree is served by a GROUP of LANES consecutive lanes of one wavefront
// (LANES = power of two >= n_actions, so a 64-action game takes the whole wave
// and a 2-action game packs up to 32 trees per wave); lane `a` of the group owns
// action `a`.  All cross-lane traffic stays inside the group.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace mz {

// Compile-time loop: f(std::integral_constant<int, 0>()), ..., f(std::integral_constant<int, N - 1>()).  Indexing
// an array in registers needs indices that are constants, which an unrolled runtime loop does not always give.
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f)
{
    (f(std::integral_constant<int, I>()), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(std::make_integer_sequence<int, N>(), f); }

// ----------------------------------------------------------------- storage layout
// One node = one contiguous record (64-byte aligned):
//   header (32 B): i32 node_visits, parent, action_from_parent, to_play; f64 node_value, node_raw_value
//   rows over actions: f64 children_values[A], f64 children_rewards[A],
//                      prior[A] (f64 for PUCT, f32 for Gumbel), children_visits[A], children_index[A]
//                      (both i32, or i16 when N < 32768 and the narrower rows save a 64-byte unit)
// Records of a tree are contiguous ([N] of them), trees are contiguous ([B]).
struct Layout {
    int A, N, H, S;
    int rec;  // bytes per node record
    int off_cval, off_crew, off_prior, off_cvis, off_cidx;
    int prior_f32;
    int idx16;  // children_visits / children_index rows stored as int16 (when that shortens the record)
    long long tree_stride;  // N * rec
};

struct NodeHdr {
    int32_t visits, parent, afp, to_play;
    double value, raw;
};

struct Dev {
    Layout L;
    char *nodes;            // [B][N][rec]
    float *hidden;          // [B][N][H]
    unsigned long long *legal;  // [B] bit a = action a legal at the root
    double *gumbel;         // [B][A]
    int32_t *m_considered;  // [B] min(max_considered, #legal)
    int32_t *sel_parent, *sel_action, *sel_ntp, *sel_depth;  // [B]
    uint32_t *noise_calls;  // [B] PUCT selection calls so far in this search
    long long *visit_total; // [B] sum over simulations of (depth + 1)
    int32_t *path;          // [B][S] (node << 6 | action) per level of the last descent
    int32_t *dec;           // [B][N] cached PUCT decisions of the level-by-level search (tree_ops.h puct_decide), or NULL
    const int32_t *cv_table;  // [(A + 1)][S] sequential-halving schedules per considered count
    const double *c_table;    // [S + 2]
    const double *sqrt_table; // [(S + 2)][S + 1] sqrt(n / (1 + c)) for small S (PUCT), else NULL
    int32_t *err;           // device error flag
    int B, players, noise_mode;
    int RL;                 // lanes of a group that hold actions: smallest power of two >= A (>= 2)
    double discount, c_visit, c_scale;
    unsigned long long seed, tree_id0;
};

// Where one tree's records and descent path live while a kernel works on it: its slices of the
// global arrays, or a copy in the block's LDS (whole-search kernels over small trees).
struct TreeMem {
    char *nodes;    // [N][rec]
    int32_t *path;  // [S]
    int32_t *depth; // [N] depth of every live node (LDS-resident trees only), else NULL
};

__device__ __forceinline__ TreeMem tree_mem(const Dev &d, int t)
{
    TreeMem m;
    m.nodes = d.nodes + (long long)t * d.L.tree_stride;
    m.path = d.path + (long long)t * d.L.S;
    m.depth = nullptr;
    return m;
}

// ----------------------------------------------------------------- group collectives
// Cross-lane movement inside a row of 16 lanes uses DPP modifiers (register-to-register, a few
// cycles); only the two widest butterfly steps (lanes 16 and 32 apart) go through ds_bpermute.
// W is the number of lanes that take part (a power of two <= LANES, uniform over the wave);
// lanes >= W of a group must not rely on the result.
#define MZ_DPP_XOR1 0xB1         // quad_perm:[1,0,3,2]
#define MZ_DPP_XOR2 0x4E         // quad_perm:[2,3,0,1]
#define MZ_DPP_HALF_MIRROR 0x141 // lane i <-> 7-i inside each 8 lanes
#define MZ_DPP_MIRROR 0x140      // lane i <-> 15-i inside each 16 lanes
#define MZ_DPP_SHL(n) (0x100 + (n))  // lane i <- lane i+n inside the row
#define MZ_DPP_SHR(n) (0x110 + (n))  // lane i <- lane i-n inside the row

template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v)
{
    // No previous value to preserve (a lane without a valid source reads 0 and no caller uses it), so this
    // is a single v_mov_b32_dpp, and for 32-bit operands the compiler folds it into the consuming op.
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, dpp_mov<CTRL>(__builtin_bit_cast(int, v)));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    long long b = __builtin_bit_cast(long long, v);
    int lo = dpp_mov<CTRL>((int)b), hi = dpp_mov<CTRL>((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// Butterfly partner of step K (pairs blocks of 2^K lanes).  Steps 2 and 3 mirror instead of xor-ing;
// after the narrower steps every block is uniform, so any lane of the partner block serves.
template <int K, typename T>
__device__ __forceinline__ T partner(T v)
{
    if constexpr (K == 0) return dpp_mov<MZ_DPP_XOR1>(v);
    else if constexpr (K == 1) return dpp_mov<MZ_DPP_XOR2>(v);
    else if constexpr (K == 2) return dpp_mov<MZ_DPP_HALF_MIRROR>(v);
    else if constexpr (K == 3) return dpp_mov<MZ_DPP_MIRROR>(v);
    else if constexpr (K == 4) return __shfl_xor(v, 16, 64);
    else return __shfl_xor(v, 32, 64);
}

template <int LANES, typename T, typename F>
__device__ __forceinline__ T greduce(T v, int W, F f)
{
    if constexpr (LANES > 1) { if (W > 1) v = f(v, partner<0>(v)); }
    if constexpr (LANES > 2) { if (W > 2) v = f(v, partner<1>(v)); }
    if constexpr (LANES > 4) { if (W > 4) v = f(v, partner<2>(v)); }
    if constexpr (LANES > 8) { if (W > 8) v = f(v, partner<3>(v)); }
    if constexpr (LANES > 16) { if (W > 16) v = f(v, partner<4>(v)); }
    if constexpr (LANES > 32) { if (W > 32) v = f(v, partner<5>(v)); }
    return v;
}

template <int LANES, typename T>
__device__ __forceinline__ T gshfl(T v, int src) { return __shfl(v, src, LANES); }

template <int LANES>
__device__ __forceinline__ double gmin(double v, int W = LANES)
{
    return greduce<LANES>(v, W, [](double x, double y) { return (y < x || y != y) ? y : x; });  // np.min: NaN wins
}
template <int LANES>
__device__ __forceinline__ double gmax(double v, int W = LANES)
{
    return greduce<LANES>(v, W, [](double x, double y) { return (y > x || y != y) ? y : x; });  // np.max: NaN wins
}
template <int LANES>
__device__ __forceinline__ float gmaxf(float v, int W = LANES)
{
    return greduce<LANES>(v, W, [](float x, float y) { return y > x ? y : x; });
}
template <int LANES>
__device__ __forceinline__ float gminf(float v, int W = LANES)
{
    return greduce<LANES>(v, W, [](float x, float y) { return y < x ? y : x; });
}
template <int LANES>
__device__ __forceinline__ int gmaxi(int v, int W = LANES)
{
    return greduce<LANES>(v, W, [](int x, int y) { return y > x ? y : x; });
}
template <int LANES>
__device__ __forceinline__ int gsumi(int v, int W = LANES)
{
    return greduce<LANES>(v, W, [](int x, int y) { return x + y; });
}
template <int LANES>
__device__ __forceinline__ float gsumf_any(float v, int W = LANES)  // order-free float sum (tolerance paths only)
{
    return greduce<LANES>(v, W, [](float x, float y) { return x + y; });
}
template <int LANES>
__device__ __forceinline__ double gsumd_any(double v, int W = LANES)
{
    return greduce<LANES>(v, W, [](double x, double y) { return x + y; });
}

// First maximal index over the group (np.argmax tie rule); `idx` must be >= A on idle lanes.
template <int LANES>
__device__ __forceinline__ int gargmax(double v, int idx, int W = LANES)
{
#define MZ_ARGMAX_STEP(K)                                         \
    {                                                             \
        double ov = partner<K>(v);                                \
        int oi = partner<K>(idx);                                 \
        bool take = (ov > v) || (ov == v && oi < idx);            \
        v = take ? ov : v;                                        \
        idx = take ? oi : idx;                                    \
    }
    if constexpr (LANES > 1) { if (W > 1) MZ_ARGMAX_STEP(0) }
    if constexpr (LANES > 2) { if (W > 2) MZ_ARGMAX_STEP(1) }
    if constexpr (LANES > 4) { if (W > 4) MZ_ARGMAX_STEP(2) }
    if constexpr (LANES > 8) { if (W > 8) MZ_ARGMAX_STEP(3) }
    if constexpr (LANES > 16) { if (W > 16) MZ_ARGMAX_STEP(4) }
    if constexpr (LANES > 32) { if (W > 32) MZ_ARGMAX_STEP(5) }
#undef MZ_ARGMAX_STEP
    return idx;
}

// Whole-wavefront reductions (a group that spans the wave): DPP only, result read back through a scalar
// register, so every lane gets it without touching LDS.  `ident` is the operation's identity.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_masked(float ident, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident),
                                                                 __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <typename F>
__device__ __forceinline__ float wave_reduce(float v, float ident, F f)
{
    v = f(v, dpp_mov<MZ_DPP_XOR1>(v));
    v = f(v, dpp_mov<MZ_DPP_XOR2>(v));
    v = f(v, dpp_mov<MZ_DPP_HALF_MIRROR>(v));
    v = f(v, dpp_mov<MZ_DPP_MIRROR>(v));                 // every lane: its row's result
    v = f(v, dpp_masked<0x142, 0xA>(ident, v));          // row_bcast:15 into rows 1 and 3
    v = f(v, dpp_masked<0x143, 0xC>(ident, v));          // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
template <typename F>
__device__ __forceinline__ double wave_reduce(double v, double ident, F f)
{
    auto mv = [](double ident_, double x, auto tag) {
        constexpr int which = decltype(tag)::value;
        long long b = __builtin_bit_cast(long long, x), ib = __builtin_bit_cast(long long, ident_);
        int lo, hi;
        if constexpr (which == 0) {
            lo = __builtin_amdgcn_update_dpp((int)ib, (int)b, 0x142, 0xA, 0xF, false);
            hi = __builtin_amdgcn_update_dpp((int)(ib >> 32), (int)(b >> 32), 0x142, 0xA, 0xF, false);
        } else {
            lo = __builtin_amdgcn_update_dpp((int)ib, (int)b, 0x143, 0xC, 0xF, false);
            hi = __builtin_amdgcn_update_dpp((int)(ib >> 32), (int)(b >> 32), 0x143, 0xC, 0xF, false);
        }
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
    };
    v = f(v, dpp_mov<MZ_DPP_XOR1>(v));
    v = f(v, dpp_mov<MZ_DPP_XOR2>(v));
    v = f(v, dpp_mov<MZ_DPP_HALF_MIRROR>(v));
    v = f(v, dpp_mov<MZ_DPP_MIRROR>(v));
    v = f(v, mv(ident, v, std::integral_constant<int, 0>()));
    v = f(v, mv(ident, v, std::integral_constant<int, 1>()));
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
// Reduction over each half of the wave (lanes 0-31 and 32-63 separately); every lane gets its half's result.
template <typename F>
__device__ __forceinline__ float half_reduce(float v, float ident, bool upper, F f)
{
    v = f(v, dpp_mov<MZ_DPP_XOR1>(v));
    v = f(v, dpp_mov<MZ_DPP_XOR2>(v));
    v = f(v, dpp_mov<MZ_DPP_HALF_MIRROR>(v));
    v = f(v, dpp_mov<MZ_DPP_MIRROR>(v));
    v = f(v, dpp_masked<0x142, 0xA>(ident, v));  // rows 1 and 3 now hold their half's result
    int b = __builtin_bit_cast(int, v);
    int lo = __builtin_amdgcn_readlane(b, 31), hi = __builtin_amdgcn_readlane(b, 63);
    return __builtin_bit_cast(float, upper ? hi : lo);
}
template <typename F>
__device__ __forceinline__ double half_reduce(double v, double ident, bool upper, F f)
{
    v = f(v, dpp_mov<MZ_DPP_XOR1>(v));
    v = f(v, dpp_mov<MZ_DPP_XOR2>(v));
    v = f(v, dpp_mov<MZ_DPP_HALF_MIRROR>(v));
    v = f(v, dpp_mov<MZ_DPP_MIRROR>(v));
    long long x = __builtin_bit_cast(long long, v), ib = __builtin_bit_cast(long long, ident);
    int l = __builtin_amdgcn_update_dpp((int)ib, (int)x, 0x142, 0xA, 0xF, false);
    int h = __builtin_amdgcn_update_dpp((int)(ib >> 32), (int)(x >> 32), 0x142, 0xA, 0xF, false);
    v = f(v, __builtin_bit_cast(double, ((long long)h << 32) | (unsigned)l));
    x = __builtin_bit_cast(long long, v);
    int lo0 = __builtin_amdgcn_readlane((int)x, 31), hi0 = __builtin_amdgcn_readlane((int)(x >> 32), 31);
    int lo1 = __builtin_amdgcn_readlane((int)x, 63), hi1 = __builtin_amdgcn_readlane((int)(x >> 32), 63);
    long long r = upper ? (((long long)hi1 << 32) | (unsigned)lo1) : (((long long)hi0 << 32) | (unsigned)lo0);
    return __builtin_bit_cast(double, r);
}
__device__ __forceinline__ double wave_lane(double v, int lane)  // lane: compile-time-constant lane index
{
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// Lane 0's value to lanes 0..W-1 of the group (W <= 16, group aligned inside a row).
template <typename T>
__device__ __forceinline__ T bcast0_row(T v, int a, int W)
{
    if (W == 2) return dpp_mov<0xA0>(v);  // quad_perm:[0,0,2,2]
    v = dpp_mov<0x00>(v);                 // quad_perm:[0,0,0,0]
    if (W > 4) { T t = dpp_mov<MZ_DPP_SHR(4)>(v); v = (a & 4) ? t : v; }
    if (W > 8) { T t = dpp_mov<MZ_DPP_SHR(8)>(v); v = (a & 8) ? t : v; }
    return v;
}

// np.sum over the A values held one per lane, in NumPy's summation order
// (numpy/_core/src/umath/loops_utils.h.src pairwise_sum: sequential below 8 elements; otherwise 8
// strided accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail in order).
// Float addition is commutative, so a butterfly over lanes 0..7 reproduces that tree exactly.
// RL = lanes holding actions (power of two, RL/2 < A <= RL, A >= 1); the result reaches all RL lanes.
// Up to 16 actions everything stays in DPP: lane 0 accumulates in order, pulling lane i with
// row_shl:i, and the sum is broadcast back.
template <int RL, typename T>
__device__ __forceinline__ T gsum_np(T x, int a, int A)
{
#define MZ_PULL(i)                                                                  \
    if constexpr ((i) < RL) {                                                       \
        if ((i) < RL / 2 || A > (i)) res += dpp_mov<MZ_DPP_SHL((i) < 16 ? (i) : 1)>(x); \
    }
    if constexpr (RL <= 4) {
        T res = x;
        MZ_PULL(1) MZ_PULL(2) MZ_PULL(3)
        return bcast0_row(res, a, RL);
    } else if constexpr (RL == 8) {
        T res;
        if (A < 8) {
            res = x;
            MZ_PULL(1) MZ_PULL(2) MZ_PULL(3) MZ_PULL(4) MZ_PULL(5) MZ_PULL(6)
        } else {
            res = x;
            res += partner<0>(res);
            res += partner<1>(res);
            res += partner<2>(res);
        }
        return bcast0_row(res, a, RL);
    } else if constexpr (RL == 16) {
        T r = x;
        if (A == 16) r += dpp_mov<MZ_DPP_SHL(8)>(x);
        r += partner<0>(r);
        r += partner<1>(r);
        r += partner<2>(r);
        T res = r;
        if (A < 16) { MZ_PULL(8) MZ_PULL(9) MZ_PULL(10) MZ_PULL(11) MZ_PULL(12) MZ_PULL(13) MZ_PULL(14) }
        return bcast0_row(res, a, RL);
    } else {
        const int lim = A - (A % 8);
        T r = x;
        for (int i = 8; i < lim; i += 8) r += __shfl(x, a + i, RL);
        r += partner<0>(r);
        r += partner<1>(r);
        r += partner<2>(r);
        T res = __shfl(r, 0, RL);
        for (int i = lim; i < A; i++) res += __shfl(x, i, RL);
        return res;
    }
#undef MZ_PULL
}

// ----------------------------------------------------------------- NumPy float32 exp
// np.exp on float32 as NumPy 2.2.6's SIMD kernel evaluates it (Cody-Waite reduction, P5/Q2
// rational, scale by 2^k).  The reference's float32 softmax (utils/utils.py:18-20 on the
// float32 prior logits of the Gumbel path and of add_exploration_noise) goes through it, and
// its results feed argmax decisions, so it is reproduced bit for bit.
__device__ __forceinline__ float np_expf(float x)
{
    const float xmax = 88.72283935546875f, xmin = -103.97208404541015625f;
    if (x != x) return x;
    if (x >= xmax) return __builtin_huge_valf();
    if (x <= xmin) return 0.0f;
    const float log2e = 0x1.715476p+0f, magic = 0x1.800000p+23f;
    float q = __fmul_rn(x, log2e);
    q = __fsub_rn(__fadd_rn(q, magic), magic);
    float r = __fmaf_rn(q, -0x1.62e400p-1f, x);
    r = __fmaf_rn(q, -0x1.7f7d1cp-20f, r);
    float num = __fmaf_rn(5.082762527590693718096e-04f, r, 6.757896990527504603057e-03f);
    num = __fmaf_rn(num, r, 5.114512081637298353406e-02f);
    num = __fmaf_rn(num, r, 2.473615434895520810817e-01f);
    num = __fmaf_rn(num, r, 7.257664613233124478488e-01f);
    num = __fmaf_rn(num, r, 9.999999999980870924916e-01f);
    float den = __fmaf_rn(2.159509375685829852307e-02f, r, -2.742335390411667452936e-01f);
    den = __fmaf_rn(den, r, 1.0f);
    return ldexpf(__fdiv_rn(num, den), (int)q);
}

// ----------------------------------------------------------------- counter noise
__device__ __host__ __forceinline__ unsigned long long splitmix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __host__ __forceinline__ unsigned long long noise_key(unsigned long long seed, unsigned long long tree)
{
    return splitmix64(seed ^ (tree * 0xD6E8FEB86659FD93ull));
}
__device__ __host__ __forceinline__ double noise_from_key(unsigned long long key, unsigned long long call,
                                                          unsigned action)
{
    unsigned long long z = splitmix64(key + call * 64 + action);
    return (double)(z >> 11) * 0x1p-53;
}
__device__ __host__ __forceinline__ double noise_u(unsigned long long seed, unsigned long long tree,
                                                   unsigned long long call, unsigned action)
{
    return noise_from_key(noise_key(seed, tree), call, action);
}

// ----------------------------------------------------------------- hash network
struct HashNet { int logit_shift, reward_on, parent_predicts; };

__device__ __host__ __forceinline__ uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __host__ __forceinline__ uint32_t hash2(uint32_t code, uint32_t k)
{
    return fmix32(code + 0x9E3779B9u * (k + 1u));
}
__device__ __forceinline__ float hash_logit(uint32_t code, int a, int shift)
{
    return ldexpf((float)((int)(hash2(code, 1 + a) % 257u) - 128), -shift);
}
__device__ __forceinline__ double hash_value(uint32_t code)
{
    return (double)((int)(hash2(code, 200) % 513u) - 256) / 256.0;
}
__device__ __forceinline__ double hash_reward(uint32_t code, int on)
{
    return on ? (double)(hash2(code, 201) % 65u) / 64.0 : 0.0;
}
__device__ __forceinline__ float hash_hidden(uint32_t code, int i)
{
    if (i == 0) return (float)(code & 0xFFFFu);
    if (i == 1) return (float)(code >> 16);
    return (float)(hash2(code, 1000 + i) & 0xFFu) / 256.0f;
}

}  // namespace mz
