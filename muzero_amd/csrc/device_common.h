// Device-side building blocks of the tree kernels (gfx950, wave64).
//
// A tree is served by a GROUP of LANES consecutive lanes of one wavefront
// (LANES = power of two >= n_actions, so a 64-action game takes the whole wave
// and a 2-action game packs up to 32 trees per wave); lane `a` of the group owns
// action `a`.  All cross-lane traffic stays inside the group.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mz {

// ----------------------------------------------------------------- storage layout
// One node = one contiguous record (64-byte aligned):
//   header (32 B): i32 node_visits, parent, action_from_parent, to_play; f64 node_value, node_raw_value
//   rows over actions: f64 children_values[A], f64 children_rewards[A],
//                      prior[A] (f64 for PUCT, f32 for Gumbel), i32 children_visits[A], i32 children_index[A]
// Records of a tree are contiguous ([N] of them), trees are contiguous ([B]).
struct Layout {
    int A, N, H, S;
    int rec;  // bytes per node record
    int off_cval, off_crew, off_prior, off_cvis, off_cidx;
    int prior_f32;
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
    const int32_t *cv_table;  // [(A + 1)][S] sequential-halving schedules per considered count
    const double *c_table;    // [S + 2]
    int32_t *err;           // device error flag
    int B, players, noise_mode;
    double discount, c_visit, c_scale;
    unsigned long long seed, tree_id0;
};

__device__ __forceinline__ char *node_rec(const Dev &d, int t, int n)
{
    return d.nodes + (long long)t * d.L.tree_stride + (long long)n * d.L.rec;
}

// ----------------------------------------------------------------- group collectives
template <int LANES, typename T>
__device__ __forceinline__ T gshfl(T v, int src) { return __shfl(v, src, LANES); }

template <int LANES>
__device__ __forceinline__ double gmin(double v)
{
#pragma unroll
    for (int m = LANES / 2; m > 0; m >>= 1) { double o = __shfl_xor(v, m, LANES); v = o < v ? o : v; }
    return v;
}
template <int LANES>
__device__ __forceinline__ double gmax(double v)
{
#pragma unroll
    for (int m = LANES / 2; m > 0; m >>= 1) { double o = __shfl_xor(v, m, LANES); v = o > v ? o : v; }
    return v;
}
template <int LANES>
__device__ __forceinline__ float gmaxf(float v)
{
#pragma unroll
    for (int m = LANES / 2; m > 0; m >>= 1) { float o = __shfl_xor(v, m, LANES); v = o > v ? o : v; }
    return v;
}
template <int LANES>
__device__ __forceinline__ int gmaxi(int v)
{
#pragma unroll
    for (int m = LANES / 2; m > 0; m >>= 1) { int o = __shfl_xor(v, m, LANES); v = o > v ? o : v; }
    return v;
}
template <int LANES>
__device__ __forceinline__ int gsumi(int v)
{
#pragma unroll
    for (int m = LANES / 2; m > 0; m >>= 1) v += __shfl_xor(v, m, LANES);
    return v;
}

// First maximal index over the group (np.argmax tie rule); `idx` must be >= A on idle lanes.
template <int LANES>
__device__ __forceinline__ int gargmax(double v, int idx)
{
#pragma unroll
    for (int m = LANES / 2; m > 0; m >>= 1) {
        double ov = __shfl_xor(v, m, LANES);
        int oi = __shfl_xor(idx, m, LANES);
        bool take = (ov > v) || (ov == v && oi < idx);
        v = take ? ov : v;
        idx = take ? oi : idx;
    }
    return idx;
}

// np.sum over the A values held one per lane, in NumPy's summation order
// (numpy/_core/src/umath/loops_utils.h.src pairwise_sum: sequential below 8 elements; otherwise 8
// strided accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail in order).
// Float addition is commutative, so the xor butterfly over lanes 0..7 reproduces that tree exactly.
template <int LANES, typename T>
__device__ __forceinline__ T gsum_np(T x, int a, int A)
{
    if (A < 8) {
        T res = (T)(-0.0);
        for (int i = 0; i < A; i++) res += gshfl<LANES>(x, i);
        return res;
    }
    if constexpr (LANES >= 8) {
        const int lim = A - (A % 8);
        T r = x;
        for (int i = 8; i < lim; i += 8) r += gshfl<LANES>(x, a + i);
        r += __shfl_xor(r, 1, LANES);
        r += __shfl_xor(r, 2, LANES);
        r += __shfl_xor(r, 4, LANES);
        T res = gshfl<LANES>(r, 0);
        for (int i = lim; i < A; i++) res += gshfl<LANES>(x, i);
        return res;
    } else {
        return x;  // unreachable: LANES >= A
    }
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
__device__ __host__ __forceinline__ double noise_u(unsigned long long seed, unsigned long long tree,
                                                   unsigned long long call, unsigned action)
{
    unsigned long long key = splitmix64(seed ^ (tree * 0xD6E8FEB86659FD93ull));
    unsigned long long z = splitmix64(key + call * 64 + action);
    return (double)(z >> 11) * 0x1p-53;
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
