// Environments stepped on the device, fused with the final action rule (one launch per self-play move).
#pragma once
#include "tree_kernels.h"

namespace mz {

// ---------------------------------------------------------------------------- self-play step
// Final action rule fused with the environment step of self_play.py:92-113 for every env of the batch:
// act, game.step(action), hand the trajectory record of this move to the caller, restart finished
// games.  Environment state is caller-owned device memory (mz_env_state in the header).
struct EnvDev {
    int kind, max_moves, stale, obs_size;
    int n_stack, planes;   // observations in the search stack, planes per observation (2 with action planes)
    void *state;
    int32_t *mover, *to_play;
    uint8_t *legal;
    float *cur_obs, *search_obs;
    int32_t *moves;
    const double *reset_state;
};

// Rows of a move log (mz_move_log); pointers NULL = no log.
struct LogDev {
    float *observations;
    int32_t *actions, *to_plays;
    double *rewards, *root_values, *probs;
    uint8_t *done;
    long long row_base;   // row * n_envs
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

// After the step: a finished game's successor starts from its first observation (followed by zero planes in a stack);
// without the stale rule every search starts from the current observation.
__device__ __forceinline__ void restart_or_refresh(const EnvDev &e, bool done, bool stacked, int SW, int OS, const float *cur,
                                                   float *srch)
{
    if (done || (!stacked && !e.stale)) for (int i = 0; i < OS; i++) srch[i] = cur[i];
    if (done && stacked) for (int j = OS; j < SW; j++) srch[j] = 0.0f;
}

template <int LANES, int RL, bool GUMBEL>
__global__ __launch_bounds__(MZ_BLOCK) void k_act_env(Dev d, EnvDev e, LogDev lg, double temperature, const double *u,
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
    int action = to_act_rule<RL, GUMBEL>(d, tree_mem(d, t), t, a, on, temperature, u ? u[t] : 0.0, &prob, &qout, &rv);
    const long long lr = lg.row_base + t;   // this environment's slot in the log row
    if (on) {
        if (out_probs) out_probs[(long long)t * A + a] = prob;
        if (lg.probs) lg.probs[lr * A + a] = prob;
    }
    if (a != 0) return;
    if (out_action) out_action[t] = action;
    if (out_root_value) out_root_value[t] = rv;
    if (out_to_play) out_to_play[t] = e.to_play[t];
    const bool stacked = e.n_stack > 1 || e.planes > 1;
    const int slot = e.planes * OS, SW = e.n_stack * slot;   // floats per stacked observation (+ action plane), per stack
    float *cur = e.cur_obs + (long long)t * OS, *srch = e.search_obs + (long long)t * SW;
    uint8_t *legal = e.legal + (long long)t * A;
    if (out_obs_before) for (int i = 0; i < OS; i++) out_obs_before[(long long)t * OS + i] = cur[i];
    if (lg.actions) {
        for (int i = 0; i < OS; i++) lg.observations[lr * OS + i] = cur[i];
        lg.actions[lr] = action; lg.to_plays[lr] = e.to_play[t]; lg.root_values[lr] = rv;
    }
    // the reference searches the NEXT move from the observation saved with this one (self_play.py:86-91,108)
    if (stacked) {
        // GameHistory.stack_n_observations(-1, ...) after this move's save (games/game.py:143-158): the saved observation
        // and its action plane in front, the older entries one slot further back; before the first save the stack held
        // the initial observation in front (games/game.py:134-141), which is this very observation and not an older one
        const bool first = e.moves[t] == 0;
        for (int j = SW - 1; j >= slot; j--) srch[j] = first ? 0.0f : srch[j - slot];
        for (int i = 0; i < OS; i++) srch[i] = cur[i];
        if (e.planes > 1) {
            const float av = (float)((double)action / (double)A);
            for (int i = 0; i < OS; i++) srch[OS + i] = av;
        }
    } else if (e.stale) {
        for (int i = 0; i < OS; i++) srch[i] = cur[i];
    }
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
        restart_or_refresh(e, done, stacked, SW, OS, cur, srch);
        if (out_done) out_done[t] = done;
        if (lg.done) lg.done[lr] = done;
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
        restart_or_refresh(e, done, stacked, SW, OS, cur, srch);
        if (out_done) out_done[t] = done;
        if (lg.done) lg.done[lr] = done;
    }
    if (out_reward) out_reward[t] = reward;
    if (lg.rewards) lg.rewards[lr] = (double)reward;
}

}  // namespace mz
