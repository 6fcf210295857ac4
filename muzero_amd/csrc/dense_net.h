// In-kernel evaluation of a lowered network (dense program, layout in muzero_amd/packing.py) by a lane
// group with weights and activations in LDS, plus the epilogue pieces every leaf evaluator shares:
// hidden-state min-max scaling and support-to-scalar.
#pragma once
#include "tree_kernels.h"

namespace mz {

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

// Whole-wave epilogue for small networks (one normalisation group of at most 16 hidden values, supports
// of at most 32 bins): the hidden-state scaling of minmax_to and both support_to_scalar chains of
// support_to_scalar2 in one branch-free stretch, so that the independent reduction chains interleave.
// Same operations per value as the two functions it replaces.
__device__ __forceinline__ void epilogue_wave_small(const float *raw, int H, const float *lgA, const float *lgB,
                                                    int n, int a, float *hid_dst, double *outA, double *outB)
{
    const float inf = __builtin_huge_valf();
    const bool up = a >= 32, has_h = a < H;
    const int i = a & 31, L = (n - 1) / 2;
    float v = i < n ? (up ? lgB : lgA)[i] : -inf;
    float r = has_h ? raw[a] : 0.0f;
    int W = 2;
    while (W < H) W <<= 1;
    float mn = gminf<16>(has_h ? r : inf, W), mx = gmaxf<16>(has_h ? r : -inf, W);
    float m = half_reduce(v, -inf, up, [](float x, float y) { return y > x ? y : x; });
    float den = (mx - mn) + 1e-5f;
    float e = i < n ? expf(v - m) : 0.0f;
    float hv = (r - mn) / den;
    float sum = half_reduce(e, 0.0f, up, [](float x, float y) { return x + y; });
    if (has_h) hid_dst[a] = hv;
    double x = i < n ? (double)(e / sum) * (double)(i - L) : 0.0;
    x = half_reduce(x, 0.0, up, [](double p, double q) { return p + q; });
    double y = inv_value_transform(x);
    *outA = wave_lane(y, 0);
    *outB = wave_lane(y, 32);
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

}  // namespace mz
