// Training-target construction over trajectories kept on the device (the caller side of the search:
// what the reference's replay buffer does with the (action, root value, policy) records the search
// produced).  One thread per (sample, unroll step); HBM-bound gathers, float64 sums in the
// reference's order, float32 outputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mz {

struct TrajDev {
    int n_games, obs_size, A;
    const long long *game_offset;  // [G + 1]
    const float *observations;     // [T][obs]
    const int32_t *actions;        // [T]
    const double *rewards;         // [T]
    const int32_t *to_plays;       // [T]
    const double *root_values, *probs;        // [T], [T][A]
    const double *re_root_values, *re_probs;  // reanalysed statistics (may be NULL)
    const uint8_t *has_re;                    // [G] (may be NULL)
    const long long *game_base;               // [G] or NULL: position i of game g at game_base[g] + i * stride
    long long stride;                         // (flat form: at game_offset[g] + i)
};

// games/game.py:205-236: value target of position `step` of a game of length T starting at `o`.
__device__ inline double value_target(const TrajDev &tr, long long o, long long st, int T, int step, int td, double gamma,
                                      const double *gamma_pow, const double *root_values)
{
    if (step >= T) return 0.0;
    const int me = tr.to_plays[o + step * st];
    double sum = 0.0;
    if (gamma == 1.0) {  // board games: signed sum of the rewards to the end of the game
        for (int i = step; i < T; i++) {
            double r = tr.rewards[o + i * st];
            sum += tr.to_plays[o + i * st] == me ? r : -r;
        }
        return sum;
    }
    const int bs = step + td;
    // rewards[step + 1 .. bootstrap_step], each signed by to_plays[step + k] (the index lags the reward's by one)
    const int last = bs + 1 < T ? bs + 1 : T;
    for (int i = step + 1, k = 0; i < last; i++, k++) {
        double r = tr.rewards[o + i * st];
        sum += (tr.to_plays[o + (step + k) * st] == me ? r : -r) * gamma_pow[k];
    }
    double bootstrap = 0.0;
    if (bs < T) {
        double v = root_values[o + bs * st];
        bootstrap = (tr.to_plays[o + bs * st] == me ? v : -v) * gamma_pow[td];
    }
    return sum + bootstrap;
}

// replay_buffer.py:70-101 for one sampled (game, t): GameHistory.stack_n_observations (games/game.py:120-160),
// the unrolled actions with absorbing padding, GameHistory.make_target (games/game.py:185-253).
__global__ void k_make_targets(TrajDev tr, int n_samples, const int32_t *game_idx, const int32_t *ts,
                               const int32_t *absorbed, int td, int unroll, double gamma, const double *gamma_pow,
                               int n_stack, int stack_action, float *out_obs, float *out_actions,
                               float *out_value, float *out_reward, float *out_policy)
{
    const int U1 = unroll + 1;
    long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)n_samples * U1) return;
    const int b = (int)(gid / U1), k = (int)(gid % U1);
    const int g = game_idx[b];
    const long long o = tr.game_base ? tr.game_base[g] : tr.game_offset[g];   // first position of the game
    const long long st = tr.game_base ? tr.stride : 1;                        // distance between its positions
    const int T = (int)(tr.game_offset[g + 1] - tr.game_offset[g]);
    const int t = ts[b], step = t + k, A = tr.A;
    const bool re = tr.has_re && tr.has_re[g];
    const double *rv = (re && tr.re_root_values) ? tr.re_root_values : tr.root_values;
    const double *pr = (re && tr.re_probs) ? tr.re_probs : tr.probs;
    double v = value_target(tr, o, st, T, step, td, gamma, gamma_pow, rv);
    float *pol = out_policy + ((long long)b * U1 + k) * A;
    if (step < T) {
        out_value[(long long)b * U1 + k] = (float)v;
        out_reward[(long long)b * U1 + k] = (float)tr.rewards[o + step * st];
        for (int a = 0; a < A; a++) pol[a] = (float)pr[(o + step * st) * A + a];
    } else {
        out_value[(long long)b * U1 + k] = 0.0f;
        out_reward[(long long)b * U1 + k] = 0.0f;
        for (int a = 0; a < A; a++) pol[a] = (float)(1.0 / (double)A);
    }
    // action taken at t + k, or a uniformly sampled earlier action of the same game past its end
    int src = step < T ? step : absorbed[(long long)b * U1 + (step - T)];
    int act = tr.actions[o + src * st];
    float *oa = out_actions + ((long long)b * U1 + k) * A;
    for (int a = 0; a < A; a++) oa[a] = a == act ? 1.0f : 0.0f;
    if (k == 0) {  // o_t, o_{t-1}, ... newest first, zero planes before the start of the game
        const int planes = stack_action ? 2 : 1, OS = tr.obs_size;
        float *oo = out_obs + (long long)b * n_stack * planes * OS;
        for (int j = 0; j < n_stack; j++) {
            const int s = t - j;
            for (int i = 0; i < OS; i++) oo[(long long)(j * planes) * OS + i] = s >= 0 ? tr.observations[(o + s * st) * OS + i] : 0.0f;
            if (stack_action) {
                float av = s >= 0 ? (float)((double)tr.actions[o + s * st] / (double)A) : 0.0f;
                for (int i = 0; i < OS; i++) oo[(long long)(j * planes + 1) * OS + i] = av;
            }
        }
    }
}

// The finished games of a move log (see mz_move_log): one thread per environment walks its end-of-game flags.
// LIST = false: count them; LIST = true: write (first position, length) of each, numbered from first_game[env].
// An environment flagged in open_at_start was mid-game at row 0: the moves up to its first flag are not a game.
template <bool LIST>
__global__ void k_log_games(const uint8_t *done, const uint8_t *open_at_start, int n_envs, int n_rows, int32_t *out_count,
                            const long long *first_game, long long *out_base, int32_t *out_len)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_envs) return;
    int n = 0, start = 0;
    long long g = LIST ? first_game[b] : 0;
    bool whole = !(open_at_start && open_at_start[b]);   // false while the walk is inside the tail of an earlier game
    for (int k = 0; k < n_rows; k++) {
        if (done[(long long)k * n_envs + b]) {
            if (whole) {
                if constexpr (LIST) { out_base[g] = (long long)start * n_envs + b; out_len[g] = k - start + 1; g++; }
                n++;
            }
            whole = true;
            start = k + 1;
        }
    }
    if constexpr (!LIST) out_count[b] = n;
}

// utils/network_utils.py:70-71 + 99-129 (scalar_to_support with the default transform): float32
// h(x) = sign(x) (sqrt(|x| + 1) - 1) + 0.001 x, clamped to [-L, L], spread over the two neighbouring
// integer bins.  One thread per output element: out[i][j], j = 0 .. 2L.
__global__ void k_scalar_to_support(const float *x, long long n, int L, float *out)
{
    const int bins = 2 * L + 1;
    long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * bins) return;
    const long long i = gid / bins;
    const int j = (int)(gid % bins);
    float v = x[i];
    float sg = v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
    float h = sg * (sqrtf(fabsf(v) + 1.0f) - 1.0f) + 0.001f * v;
    h = fminf(fmaxf(h, (float)-L), (float)L);
    float fl = floorf(h);
    float prob = h - fl;
    int lo = (int)fl + L;
    float r = 0.0f;
    if (j == lo) r = 1.0f - prob;
    else if (j == lo + 1) r = prob;   // lo + 1 > 2L only when prob == 0: nothing to write
    out[gid] = r;
}

}  // namespace mz
