// Whole PUCT search of a 2-action tree of at most 64 nodes by ONE wavefront with the tree in registers:
// lane j owns node j (header fields and both actions' statistics, ~25 registers), so
//  * selection (mcts/action_selection.py:25-49 with mcts/qtransforms.py:7-22) is evaluated at every node at
//    once without a single memory access or cross-lane step - both actions of a node sit in the same lane -
//    and the descent (mcts/mcts.py:98-114) is read off with one lane read per level;
//  * the backup (mcts/mcts.py:143-164) updates every node of the path in its own lane; only the discounted
//    return, a serial recurrence over the path's rewards, walks the levels (lane reads), and the child's
//    new mean reaches its parent's lane with one cross-lane read;
//  * expansion (mcts/mcts.py:116-141) is a predicated register write in the leaf's and the parent's lanes.
// Small batches are bound by the latency of one wave's dependent instructions and LDS round trips, not by
// bandwidth; this form removes nearly all LDS traffic of the tree.  The network is evaluated as in
// k_search_net (register-cached weights, wave epilogue) with hidden states in LDS, or, for the smallest
// networks (TINY), with the hidden states in registers too: the parent's state reaches the first level as
// wave-uniform operands (lane reads), the second level's rows are placed on the lanes where the epilogue
// wants their outputs, and only the hand-over between the two levels goes through LDS.  Every expression is
// the one the lane-group kernels evaluate (tree_ops.h, dense_net.h), so trees are identical bit for bit.
#pragma once
#include "search_net.h"

namespace mz {

__device__ __forceinline__ double lane_read(double v, int lane)  // lane: wave-uniform
{
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// One tree per wave; the block may hold several waves.  Dynamic LDS: weights | program | per-wave
// activation scratch | per-wave hidden states [N][H] (`hid_bytes` each, 16-byte aligned).
// Requires (checked by the host): PUCT, A == 2, N <= 64, a program dense_run_cached serves, one
// normalisation group of at most 16 hidden values and supports of at most 32 bins.
// TINY (HT = hidden size, 4 or 8; 0 = not TINY) additionally: two levels without residual inputs, every
// first-level row reads the network input (hidden state followed by the 2-value one-hot action), supports
// of at most 24 bins and the second level produces exactly the four outputs (then no LDS is reserved for
// hidden states).  The hidden size is a template parameter so that the input vector's layout is static.
// Lane placement of the second level's rows under TINY: value logit i on lane i, next hidden state k on
// lane 24 + k, reward logit i on lane 32 + i, policy logit a on lane 56 + a.
#define MZ_TINY_HID 24
#define MZ_TINY_POL 56
// The final action rule (k_to_act) can run at the end of the same launch: `act.action` non-NULL.
struct ActOut {
    double temperature;
    const double *u;       // [B] uniforms of the final sampling, or NULL
    int32_t *action;       // [B], NULL: no final rule here
    double *probs, *root_value, *q;  // [B][A], [B], [B][A] (q may be NULL)
};

// HLDS (TINY only): hidden states in LDS instead of registers - one broadcast read feeds the first level and one
// store keeps the new state, 22 vector instructions fewer per simulation for two more LDS round trips: the better
// trade when the chip is full and instruction issue is the bound, the worse one for a lone wave per SIMD.
template <int HT, bool HLDS = false>
__global__ __launch_bounds__(MZ_BLOCK) void k_search_reg(Dev d, NetDev net, int first_leaf, int n_sims,
                                                          const double *noise, int noise_max_calls,
                                                          int scratch_floats, int hid_bytes, ActOut act)
{
    extern __shared__ __align__(16) float lds[];
    float *w = lds;
    int32_t *meta = reinterpret_cast<int32_t *>(lds + net.n_data);
    float *scratch0 = lds + net.n_data + net.n_meta;
    constexpr bool TINY = HT > 0;
    const int n_threads = blockDim.x, waves_per_block = n_threads / 64;
    const int wv = threadIdx.x / 64, lane = threadIdx.x % 64;
    for (int i = threadIdx.x; i < net.n_data; i += n_threads) w[i] = net.data[i];
    for (int i = threadIdx.x; i < net.n_meta; i += n_threads) meta[i] = net.meta[i];
    float *sc = scratch0 + (size_t)wv * scratch_floats;
    for (int i = lane; i < scratch_floats; i += 64) sc[i] = 0.0f;
    __shared__ __align__(16) double s_ct[MZ_CT_LDS];
    const double *ct = stage_c_table<false>(d, s_ct);
    __shared__ __align__(16) double s_sq[MZ_SQ_LDS];
    const double *sqt = stage_sqrt_table<false>(d, s_sq);
    __syncthreads();
    const int t = (int)blockIdx.x * waves_per_block + wv;
    if (t >= d.B) return;
    const int H = TINY ? HT : d.L.H, S = d.L.S;
    size_t base = ((size_t)(net.n_data + net.n_meta) + (size_t)waves_per_block * scratch_floats) * 4;
    base = (base + 15) & ~(size_t)15;
    float *hid = reinterpret_cast<float *>(reinterpret_cast<char *>(lds) + base + (size_t)wv * hid_bytes);
    float *const hid_global = d.hidden + (long long)t * d.L.N * H;
    const TreeMem mg = tree_mem(d, t);
    constexpr int HR = TINY ? HT : 1;
    float hreg[HR];  // TINY: this lane's node's hidden state
#pragma unroll
    for (int k = 0; k < HR; k++) hreg[k] = 0.0f;
    if constexpr (TINY && !HLDS) {
        if (lane < first_leaf) {
#pragma unroll
            for (int k = 0; k < HT; k++) hreg[k] = hid_global[lane * HT + k];
        }
    } else {
        for (int i = lane; i < first_leaf * H; i += 64) hid[i] = hid_global[i];
    }

    // ---- this lane's node
    int nv = 0, par = -1, afp = -1, tp = 0, dep = 0;
    double nval = 0.0, raw = 0.0;
    int cv0 = 0, cv1 = 0, ci0 = -1, ci1 = -1;
    double pr0 = 0.0, pr1 = 0.0, r0 = 0.0, r1 = 0.0, v0 = 0.0, v1 = 0.0;
    if (lane < first_leaf) {
        const RowView rv = view(d, mg, lane);
        const NodeHdr h = *rv.hdr;
        nv = h.visits; par = h.parent; afp = h.afp; tp = h.to_play; nval = h.value; raw = h.raw;
        cv0 = rv.cvis.get(0); cv1 = rv.cvis.get(1); ci0 = rv.cidx.get(0); ci1 = rv.cidx.get(1);
        const double *pp = reinterpret_cast<const double *>(rv.prior);
        pr0 = pp[0]; pr1 = pp[1]; r0 = rv.crew[0]; r1 = rv.crew[1]; v0 = rv.cval[0]; v1 = rv.cval[1];
    }
    int maxdep = 0;  // depth of the deepest node so far (wave-uniform)
    for (int n = 1; n < first_leaf; n++) {  // parents precede their children: one forward sweep gives every depth
        const int p = __builtin_amdgcn_readlane(par, n);
        const int dp = __builtin_amdgcn_readlane(dep, p) + 1;
        dep = lane == n ? dp : dep;
        maxdep = dp > maxdep ? dp : maxdep;
    }

    const double inf = __builtin_huge_val();
    const unsigned long long key = noise_key(d.seed, d.tree_id0 + (unsigned long long)t);
    const unsigned long long legal = d.legal[t];
    const bool legal0 = lane != 0 || (legal & 1ull), legal1 = lane != 0 || ((legal >> 1) & 1ull);
    const double *nz = noise ? noise + (long long)t * noise_max_calls * 2 : nullptr;
    uint32_t calls = d.noise_calls[t];
    uint32_t wbase = calls - 64u;  // noise window: empty
    double w0 = 0.0, w1 = 0.0;
    long long visited = 0;
    const int asz = net.io[8], sup = net.io[9];
    DenseRegs<2, 4> regs;
    dense_cache_load<64, 2, 4>(meta, w, lane, regs);
    if constexpr (TINY) {
        // second level: this lane takes the row whose output the epilogue wants here (or none)
        const int32_t *lv = meta + meta[5], *rows = meta + meta[6];
        const int n_rows2 = lv[4], row0 = lv[7];
        const float4 *W4 = reinterpret_cast<const float4 *>(w + lv[6]);
        int tgt = -1;
        if (lane < sup) tgt = net.io[5] + lane;
        else if (lane >= MZ_TINY_HID && lane < MZ_TINY_HID + H) tgt = net.io[2] + lane - MZ_TINY_HID;
        else if (lane >= 32 && lane < 32 + sup) tgt = net.io[3] + lane - 32;
        else if (lane >= MZ_TINY_POL && lane < MZ_TINY_POL + 2) tgt = net.io[4] + lane - MZ_TINY_POL;
        int mine = -1;
        for (int r = 0; r < n_rows2; r++)
            if (rows[4 * (row0 + r) + 1] == tgt) mine = r;
        regs.ds[1] = make_int4(0, -1, 0, -1);
        regs.bias[1] = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) regs.w[1][k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mine >= 0) {
            regs.ds[1] = *reinterpret_cast<const int4 *>(rows + 4 * (row0 + mine));
            regs.bias[1] = (w + meta[4])[row0 + mine];
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k < regs.k4n[1]) regs.w[1][k] = W4[k * n_rows2 + mine];
        }
    }
    lds_fence();
#ifdef MZ_STAMPS
    long long st_sel = 0, st_net = 0, st_epi = 0, st_bk = 0, st_in = 0, st_mm = 0, st_t;
    st_t = clock64();
#endif
    for (int s = 0; s < n_sims; s++) {
        const int leaf = first_leaf + s;
        // ---- selection at every node (this lane: its node, both actions)
        // tie-break noise of selection call (calls + depth), both actions: the same for every node of a depth.
        // Lane l keeps the draws of call (wbase + l); the window is redrawn when this simulation could run past it
        // (a simulation uses up `depth` calls, so one window lasts about ten simulations).
        double u0 = 0.0, u1 = 0.0;
        if (d.noise_mode == 0) {
            if ((int)(calls - wbase) + maxdep > 63) {
                asm volatile("; redraw the noise window" ::: "memory");  // not to be evaluated speculatively
                wbase = calls;
                w0 = noise_from_key(key, wbase + (uint32_t)lane, 0u);
                w1 = noise_from_key(key, wbase + (uint32_t)lane, 1u);
            }
            const int src = (int)(calls - wbase) + dep;
            u0 = __shfl(w0, src, 64);
            u1 = __shfl(w1, src, 64);
        } else if (d.noise_mode == 1) {
            const uint32_t call = calls + (uint32_t)dep;
            if ((int)call < noise_max_calls) {
                u0 = nz[(long long)call * 2];
                u1 = nz[(long long)call * 2 + 1];
            }
        }
        const double q0 = r0 + d.discount * v0, q1 = r1 + d.discount * v1;
        const double s0 = cv0 > 0 ? q0 : nval, s1 = cv1 > 0 ? q1 : nval;
        // min / max over the two actions (np.min / np.max: a NaN wins).  The 2-lane groups compute them per action
        // lane with the operands swapped, which can only differ in the sign of a zero when the two compare equal -
        // and that never reaches a decision: (cq - lo) is then +0 for the tied action either way
        double lo0 = (s1 < s0 || s1 != s1) ? s1 : s0;
        double hi0 = (s1 > s0 || s1 != s1) ? s1 : s0;
        lo0 = nval < lo0 ? nval : lo0;
        hi0 = nval > hi0 ? nval : hi0;
        double den0 = hi0 - lo0;
        den0 = den0 < 1e-8 ? 1e-8 : den0;
        const double lo1 = lo0, den1 = den0;
        const double cq0 = cv0 > 0 ? q0 : lo0, cq1 = cv1 > 0 ? q1 : lo1;
        double cc;
        if (ct) {
            cc = ct[nv];  // LDS
        } else {
            asm volatile("; exploration constants from global memory" ::: "memory");  // two typed loads, not one generic
            cc = d.c_table[nv];
        }
        const double sq0 = sqt ? sqt[nv * (S + 1) + cv0] : sqrt((double)nv / (double)(1 + cv0));
        const double sq1 = sqt ? sqt[nv * (S + 1) + cv1] : sqrt((double)nv / (double)(1 + cv1));
        const double uu0 = cc * pr0 + sq0, uu1 = cc * pr1 + sq1;
        // The rule needs the larger of two scores, not the scores.  Both transformed values divide by the same range:
        // with its reciprocal (two Newton steps: within 3e-16 of the quotient's factor) each is within 4e-16 of the
        // correctly rounded division, the sums that follow add a few ulp of their operands, so the cheap scores are
        // within 1e-14 * (1 + |uu0| + |uu1|) of the reference's, and their order is the reference's whenever they
        // differ by more than that (the tie-break term alone, 1e-7 * |u0 - u1|, is usually 10^5 times larger).  Only
        // the remaining cases - and NaN or doubly illegal scores, whose difference is NaN - divide.
        double rden = __builtin_amdgcn_rcp(den0);
        rden = __builtin_fma(__builtin_fma(-den0, rden, 1.0), rden, rden);
        rden = __builtin_fma(__builtin_fma(-den0, rden, 1.0), rden, rden);
        double f0 = (uu0 + (cq0 - lo0) * rden) + 1e-7 * u0, f1 = (uu1 + (cq1 - lo1) * rden) + 1e-7 * u1;
        if (!legal0) f0 = -inf;
        if (!legal1) f1 = -inf;
        const double lead = f1 - f0;
        int best = lead > 0.0 ? 1 : 0;
        if (!(__builtin_fabs(lead) > 1e-14 * (1.0 + __builtin_fabs(uu0) + __builtin_fabs(uu1)))) {
            const double nq0 = (cq0 - lo0) / den0, nq1 = (cq1 - lo1) / den1;
            double sc0 = (uu0 + nq0) + 1e-7 * u0, sc1 = (uu1 + nq1) + 1e-7 * u1;
            sc0 = sc0 != sc0 ? inf : sc0; sc1 = sc1 != sc1 ? inf : sc1;  // np.argmax: a NaN is the maximum
            if (!legal0) sc0 = -inf;
            if (!legal1) sc1 = -inf;
            best = sc1 > sc0 ? 1 : 0;  // first maximum
        }
        const int chosen_child = best ? ci1 : ci0;
        const int pack = ((chosen_child + 1) << 8) | best;
        // ---- descent: follow the decisions from the root
        int node = 0, depth = 0, pk;
        unsigned long long onpath = 0ull;
        while (true) {
            pk = __builtin_amdgcn_readlane(pack, node);
            onpath |= 1ull << node;
            depth++;
            const unsigned child = (unsigned)(pk >> 8) - 1u;  // no child: 0xffffffff
            if (child >= (unsigned)leaf || depth > S) break;  // depth bound: a malformed tree cannot spin forever
            node = (int)child;
        }
        const int action = pk & 0xff;
        const int ntp = d.players == 1 ? 0 : (depth & 1);  // starts at player 0, flips once per level (mcts/mcts.py:105,112)
        const int parent = node;
        calls += (uint32_t)depth;
        visited += depth + 1;
#ifdef MZ_STAMPS
        MZ_STAMP(st_sel)
#endif
        // ---- leaf evaluation: the PARENT's hidden state and the encoded action (mcts/mcts.py:240-243)
        double value, reward;
        float lg0, lg1;
        if constexpr (TINY) {
            // network input as wave-uniform values: x[k] = hidden[k] (k < HT), then the one-hot action, then zeros
            float x[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k < HT) {
                    if constexpr (HLDS) x[k] = hid[parent * HT + k];  // same address on every lane: a broadcast read
                    else x[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hreg[k < HT ? k : 0]), parent));
                } else {
                    x[k] = (k - HT < 2 && k - HT == action) ? 1.0f : 0.0f;
                }
            }
#ifdef MZ_STAMPS
            MZ_STAMP(st_in)
#endif
            {   // first level: this lane's row, same multiply-add order as dense_run
                const int4 ds = regs.ds[0];
                float acc = regs.bias[0];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k < regs.k4n[0]) {
                        const float4 wv = regs.w[0][k];
                        acc = fmaf(wv.x, x[4 * k], acc);
                        acc = fmaf(wv.y, x[4 * k + 1], acc);
                        acc = fmaf(wv.z, x[4 * k + 2], acc);
                        acc = fmaf(wv.w, x[4 * k + 3], acc);
                    }
                }
                if (ds.z == 1) acc = acc > 0.0f ? acc : 0.0f;
                else if (ds.z == 2) acc = acc > 0.0f ? acc : expm1f(acc);
                if (ds.y >= 0) sc[ds.y] = acc;
            }
            lds_fence();
            float out2 = 0.0f;
            {   // second level: output stays in this lane
                const int4 ds = regs.ds[1];
                if (ds.y >= 0) {
                    const float4 *x4 = reinterpret_cast<const float4 *>(sc + ds.x);
                    float acc = regs.bias[1];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (k < regs.k4n[1]) {
                            const float4 wv = regs.w[1][k], xv = x4[k];
                            acc = fmaf(wv.x, xv.x, acc);
                            acc = fmaf(wv.y, xv.y, acc);
                            acc = fmaf(wv.z, xv.z, acc);
                            acc = fmaf(wv.w, xv.w, acc);
                        }
                    }
                    if (ds.z == 1) acc = acc > 0.0f ? acc : 0.0f;
                    else if (ds.z == 2) acc = acc > 0.0f ? acc : expm1f(acc);
                    out2 = acc;
                }
            }
            lds_fence();
#ifdef MZ_STAMPS
            MZ_STAMP(st_net)
#endif
            // epilogue_wave_small on registers: value logits on lanes 0.., reward logits on lanes 32.., the raw
            // next hidden state on lanes MZ_TINY_HID.. (min / max are exact, so the lanes they sit on do not matter)
            const float finf = __builtin_huge_valf();
            const bool up = lane >= 32, has_h = lane >= MZ_TINY_HID && lane < MZ_TINY_HID + H;
            const int i = lane & 31, Ls = (sup - 1) / 2;
            const float v = i < sup ? out2 : -finf;
            const float r = has_h ? out2 : 0.0f;
            constexpr int W = HT <= 2 ? 2 : HT <= 4 ? 4 : 8;
            const float mn = gminf<16>(has_h ? r : finf, W), mx = gmaxf<16>(has_h ? r : -finf, W);
            const float m = half_reduce(v, -finf, up, [](float p, float q) { return q > p ? q : p; });
            const float den = (mx - mn) + 1e-5f;
            const float e = i < sup ? expf(v - m) : 0.0f;
            const float hv = (r - mn) / den;
            const float sum = half_reduce(e, 0.0f, up, [](float p, float q) { return p + q; });
            double xs = i < sup ? (double)(e / sum) * (double)(i - Ls) : 0.0;
            xs = half_reduce(xs, 0.0, up, [](double p, double q) { return p + q; });
            const double y = inv_value_transform(xs);
            value = wave_lane(y, 0);
            reward = wave_lane(y, 32);
            if constexpr (HLDS) {
                if (has_h) hid[leaf * HT + lane - MZ_TINY_HID] = hv;
            } else {
#pragma unroll
                for (int k = 0; k < HT; k++) {
                    const float hk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv), MZ_TINY_HID + k));
                    hreg[k] = lane == leaf ? hk : hreg[k];
                }
            }
            lg0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, out2), MZ_TINY_POL));
            lg1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, out2), MZ_TINY_POL + 1));
        } else {
            const float *ph = hid + parent * H;
            for (int i = lane; i < H; i += 64) sc[net.io[0] + i] = ph[i];
            for (int i = lane; i < asz; i += 64) sc[net.io[1] + i] = i == action ? 1.0f : 0.0f;
            lds_fence();
#ifdef MZ_STAMPS
            MZ_STAMP(st_in)
#endif
            dense_run_cached<2, 4>(regs, sc);
#ifdef MZ_STAMPS
            MZ_STAMP(st_net)
#endif
            epilogue_wave_small(sc + net.io[2], H, sc + net.io[5], sc + net.io[3], sup, lane, hid + leaf * H, &value, &reward);
            lg0 = sc[net.io[4]];
            lg1 = sc[net.io[4] + 1];
        }
#ifdef MZ_STAMPS
        MZ_STAMP(st_mm)
#endif
        // ---- backup.  Level i above the leaf is the path's node at depth (depth - 1 - i); its return is
        // L_i with L_0 = value, L_{i+1} = reward_i + gamma L_i (reward_0 = the new edge's reward).
        double L = value, myL = 0.0;
        const double rsel = best ? r1 : r0;  // reward of the edge this node's decision follows
        for (int i = 0, nd = parent; i < depth; i++) {
            myL = lane == nd ? L : myL;
            const double ri = i == 0 ? reward : lane_read(rsel, nd);
            L = ri + d.discount * L;
            nd = __builtin_amdgcn_readlane(par, nd);  // up one level (the root's -1 ends the walk with the loop)
        }
        const bool onp = (onpath >> lane) & 1ull;
        const double newV = (nval * (double)nv + (tp == ntp ? myL : -myL)) / (double)(nv + 1);
        const double below = __shfl(newV, chosen_child & 63, 64);  // the path child's new mean (its lane = its index)
        const double child_new = lane == parent ? value : below;
        if (onp) {
            nv += 1;
            nval = newV;
            if (best) { v1 = child_new; cv1 += 1; } else { v0 = child_new; cv0 += 1; }
        }
        // ---- expansion
        if (lane == parent) {
            if (action) { ci1 = leaf; r1 = reward; } else { ci0 = leaf; r0 = reward; }
        }
        if (lane == leaf) {  // the lane still holds a fresh node's child rows (no visits, no links, zero rewards and values)
            nv = 1; par = parent; afp = action; tp = ntp; nval = value; raw = value; dep = depth;
            pr0 = (double)lg0; pr1 = (double)lg1;
        }
        maxdep = depth > maxdep ? depth : maxdep;
        lds_fence();
#ifdef MZ_STAMPS
        MZ_STAMP(st_bk)
#endif
    }
    if (lane == 0) {
        store_noise_calls(d, t, calls, noise_max_calls);
        d.visit_total[t] += visited;
    }
    // ---- write the tree back
    const int live = first_leaf + n_sims;
    if (lane < live) {
        const RowView rv = view(d, mg, lane);
        NodeHdr h;
        h.visits = nv; h.parent = par; h.afp = afp; h.to_play = tp; h.value = nval; h.raw = raw;
        *rv.hdr = h;
        rv.cvis.set(0, cv0); rv.cvis.set(1, cv1); rv.cidx.set(0, ci0); rv.cidx.set(1, ci1);
        double *pp = reinterpret_cast<double *>(rv.prior);
        pp[0] = pr0; pp[1] = pr1; rv.crew[0] = r0; rv.crew[1] = r1; rv.cval[0] = v0; rv.cval[1] = v1;
    }
    if constexpr (TINY && !HLDS) {
        if (lane < live) {
#pragma unroll
            for (int k = 0; k < HT; k++) hid_global[lane * HT + k] = hreg[k];
        }
    } else {
        for (int i = lane; i < live * H; i += 64) hid_global[i] = hid[i];
    }
    if (act.action) {
        // the root as a record in LDS (the activation scratch is free now); every pair of lanes then evaluates
        // k_to_act's 2-lane rule on it, lanes 0 and 1 store
        lds_fence();
        const TreeMem ml{reinterpret_cast<char *>(sc), nullptr, nullptr};
        if (lane == 0) {
            const RowView rv = view(d, ml, 0);
            NodeHdr h;
            h.visits = nv; h.parent = par; h.afp = afp; h.to_play = tp; h.value = nval; h.raw = raw;
            *rv.hdr = h;
            rv.cvis.set(0, cv0); rv.cvis.set(1, cv1); rv.cidx.set(0, ci0); rv.cidx.set(1, ci1);
            double *pp = reinterpret_cast<double *>(rv.prior);
            pp[0] = pr0; pp[1] = pr1; rv.crew[0] = r0; rv.crew[1] = r1; rv.cval[0] = v0; rv.cval[1] = v1;
        }
        lds_fence();
        const bool on = lane < 2;
        double prob, qout, rvalue;
        const int chosen = to_act_rule<2, false>(d, ml, t, lane & 1, true, act.temperature, act.u ? act.u[t] : 0.0,
                                                 &prob, &qout, &rvalue);
        if (on) {
            act.probs[(long long)t * 2 + lane] = prob;
            if (act.q) act.q[(long long)t * 2 + lane] = qout;
        }
        if (lane == 0) { act.action[t] = chosen; act.root_value[t] = rvalue; }
    }
#ifdef MZ_STAMPS
    if (lane == 0) {
        long long *o = reinterpret_cast<long long *>(d.path + (long long)t * d.L.S);
        o[0] = st_sel; o[1] = st_net; o[2] = st_epi; o[3] = st_bk; o[4] = st_in; o[5] = st_mm;
    }
#endif
}

}  // namespace mz
