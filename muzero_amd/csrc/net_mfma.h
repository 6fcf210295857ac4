// Batched leaf evaluation on the matrix cores for networks whose weights do not fit in LDS.
//
// One workgroup (4 waves) evaluates the lowered network (dense program, MFMA layout of
// muzero_amd/packing.py) for a tile of TM = 32 trees.  The tile's activations live in LDS
// ([32][stride] floats, stride == 4 mod 64 so the 16-byte operand reads are conflict-free); weights
// stream from L2 in v_mfma_f32_16x16x4_f32 operand order, one 16-byte load per lane per four k-steps, and every loaded fragment is used for both 16-row halves
// of the tile (two independent accumulators, which also covers the MFMA's dependent latency).
// Within a level the (step, column tile) jobs are dealt round-robin to the four waves; levels are
// separated by a workgroup barrier.  float32 in, float32 accumulate: a k-ordered fma chain per
// output, same rounding class as the reference's float32 network.
#pragma once
#include "tree_ops.h"

namespace mz {

#define MZ_NET_TM 32
#define MZ_NET_NT 512  // 8 waves: two per SIMD, so one wave's weight-load and LDS waits hide behind the other's MFMAs

struct MfmaNet {
    const int32_t *steps;        // 12 ints per step
    const int32_t *tiles;        // (step, column tile) pairs
    const int32_t *level_start;  // [n_levels + 1]
    const float *wfrag;
    const float *bias;
    int n_steps, n_levels, stride;
    int io[12];  // in_hidden, in_action, out_next, out_reward, out_policy, out_value, H, A, action_size, support, norm_groups, obs
};

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float apply_act(float v, int act)
{
    if (act == 1) return v > 0.0f ? v : 0.0f;
    if (act == 2) return v > 0.0f ? v : expm1f(v);
    return v;
}

// One (step, column tile) job for both 16-row halves of the tile.  Lane (r = lane & 15, q = lane >> 4)
// supplies, for the j-th of four k-steps of a 16-wide k block, A[row r][k = 16*blk + 4q + j] (four
// consecutive floats of its tree's activations: one 16-byte LDS read) and the matching weight
// W[col][same k] (one 16-byte global read of the pre-swizzled fragment).
#define MZ_WPF 10  // weight fragments prefetched per job (k blocks of 16; larger K runs in rounds of this)
__device__ __forceinline__ void mfma_tile(const MfmaNet &net, float *act, const int32_t *st, int ct, int lane)
{
    const int N = st[1], k16n = st[2], in_off = st[3], out_off = st[4], skip_off = st[5], actf = st[6];
    const float4 *wp = reinterpret_cast<const float4 *>(net.wfrag + st[7]) + (long long)ct * k16n * 64 + lane;
    const float *a0 = act + (lane & 15) * net.stride + in_off + (lane >> 4) * 4;
    const float *a1 = a0 + 16 * net.stride;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int k0 = 0; k0 < k16n; k0 += MZ_WPF) {
        float4 w[MZ_WPF];
#pragma unroll
        for (int i = 0; i < MZ_WPF; i++)
            if (k0 + i < k16n) w[i] = wp[(k0 + i) * 64];
#pragma unroll
        for (int i = 0; i < MZ_WPF; i++) {
            if (k0 + i < k16n) {
                float4 x0 = *reinterpret_cast<const float4 *>(a0 + (k0 + i) * 16);
                float4 x1 = *reinterpret_cast<const float4 *>(a1 + (k0 + i) * 16);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, w[i].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, w[i].x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, w[i].y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, w[i].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.z, w[i].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, w[i].z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.w, w[i].w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, w[i].w, acc1, 0, 0, 0);
            }
        }
    }
    // C/D layout of 16x16x4: column = lane & 15, row = (lane >> 4) * 4 + register
    const int col = ct * 16 + (lane & 15);
    if (col < N) {
        const float b = net.bias[st[8] + col];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int r0 = (lane >> 4) * 4 + j;
            float *o0 = act + r0 * net.stride, *o1 = o0 + 16 * net.stride;
            float v0 = acc0[j] + b, v1 = acc1[j] + b;
            if (skip_off >= 0) { v0 += o0[skip_off + col]; v1 += o1[skip_off + col]; }
            o0[out_off + col] = apply_act(v0, actf);
            o1[out_off + col] = apply_act(v1, actf);
        }
    }
}

// recurrent_inference (network.py:105-126) for every tree's pending selection, then nothing else:
// outputs go to the leaf's hidden slot and to the per-tree logits/value/reward buffers that
// k_expand_backup consumes.
__global__ __launch_bounds__(MZ_NET_NT) void k_net_step(Dev d, MfmaNet net, int leaf, float *out_logits,
                                                   double *out_value, double *out_reward)
{
    extern __shared__ __align__(16) float act[];  // [MZ_NET_TM][stride]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * MZ_NET_TM;
    const int H = net.io[6], A = net.io[7], asz = net.io[8], sup = net.io[9], ngroups = net.io[10];
    // Zero what the program reads but never writes: the padding behind every tensor (weights are zero
    // there, but 0 * garbage must stay finite).  Cheapest is the whole tile, as 16-byte stores.
    {
        float4 z = {0, 0, 0, 0};
        float4 *a4 = reinterpret_cast<float4 *>(act);
        for (int i = tid; i < MZ_NET_TM * net.stride / 4; i += MZ_NET_NT) a4[i] = z;
    }
    __syncthreads();
    // inputs: hidden state of each tree's selected PARENT and the encoded action (mcts/mcts.py:240-243);
    // 16 threads per tree so all 32 trees' dependent loads (parent index -> hidden row) are in flight together
    {
        constexpr int PT = MZ_NET_NT / MZ_NET_TM;
        const int i = tid / PT, part = tid % PT, t = t0 + i;
        if (t < d.B) {
            int parent = d.sel_parent[t], action = d.sel_action[t];
            const float *ph = d.hidden + ((long long)t * d.L.N + parent) * H;
            float *row = act + i * net.stride;
            if ((H & 3) == 0) {
                const float4 *p4 = reinterpret_cast<const float4 *>(ph);
                float4 *r4 = reinterpret_cast<float4 *>(row + net.io[0]);
                for (int j = part; j < H / 4; j += PT) r4[j] = p4[j];
            } else {
                for (int j = part; j < H; j += PT) row[net.io[0] + j] = ph[j];
            }
            for (int j = part; j < asz; j += PT) row[net.io[1] + j] = j == action ? 1.0f : 0.0f;
        }
    }
    __syncthreads();
    for (int l = 0; l < net.n_levels; l++) {
        for (int ti = net.level_start[l] + wave; ti < net.level_start[l + 1]; ti += MZ_NET_NT / 64) {
            int s = net.tiles[2 * ti], ct = net.tiles[2 * ti + 1];
            mfma_tile(net, act, net.steps + 12 * s, ct, lane);
        }
        __syncthreads();
    }
    // next hidden state: per-group min-max scaling (utils/network_utils.py:132-158) into the leaf slot
    const int per = H / ngroups;
    for (int job = tid; job < MZ_NET_TM * ngroups; job += MZ_NET_NT) {
        int i = job / ngroups, g = job % ngroups, t = t0 + i;
        if (t >= d.B) continue;
        const float *raw = act + i * net.stride + net.io[2] + g * per;
        float mn = raw[0], mx = raw[0];
        for (int j = 1; j < per; j++) { float v = raw[j]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        float den = (mx - mn) + 1e-5f;
        float *dst = d.hidden + ((long long)t * d.L.N + leaf) * H + g * per;
        for (int j = 0; j < per; j++) dst[j] = (raw[j] - mn) / den;
    }
    for (int job = tid; job < MZ_NET_TM * A; job += MZ_NET_NT) {
        int i = job / A, a = job % A, t = t0 + i;
        if (t < d.B) out_logits[(long long)t * A + a] = act[i * net.stride + net.io[4] + a];
    }
    // value and reward: softmax over the support (float32), expectation in float64, inverse transform
    for (int job = tid; job < MZ_NET_TM * 2; job += MZ_NET_NT) {
        int i = job >> 1, which = job & 1, t = t0 + i;
        if (t >= d.B) continue;
        const float *lg = act + i * net.stride + (which ? net.io[3] : net.io[5]);
        const int L = (sup - 1) / 2;
        float mx = lg[0];
        for (int j = 1; j < sup; j++) mx = lg[j] > mx ? lg[j] : mx;
        float s = 0.0f;
        for (int j = 0; j < sup; j++) s += expf(lg[j] - mx);
        double x = 0.0;
        for (int j = 0; j < sup; j++) x += (double)(expf(lg[j] - mx) / s) * (double)(j - L);
        const double eps = 0.001;
        double sg = x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0);
        double tt = (sqrt(1 + 4 * eps * (fabs(x) + 1 + eps)) - 1) / (2 * eps);
        double y = sg * (tt * tt - 1);
        if (which) out_reward[t] = y; else out_value[t] = y;
    }
}

}  // namespace mz
