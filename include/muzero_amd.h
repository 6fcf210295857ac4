/*
 * muzero_amd.h -- C ABI of the MI355X (gfx950) batched MuZero search engine.
 *
 * Shared library: muzero_amd/libmuzero_amd.so.  Plain pointers and sizes only.
 * Unless a parameter is suffixed _h, pointers are DEVICE pointers and the work
 * is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream).
 * Every call returns 0 on success and a negative code on error; the message
 * for the calling thread's last error is mz_last_error().
 *
 * Each entry point names the reference code (trunghng/muzero, path:line) whose
 * behaviour it replaces for a batch of B independent trees.
 */
#ifndef MUZERO_AMD_H
#define MUZERO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZ_OK 0
#define MZ_ERR_ARG (-1)      /* invalid argument / shape */
#define MZ_ERR_HIP (-2)      /* HIP runtime failure */
#define MZ_ERR_STATE (-3)    /* call out of order (e.g. expand before select) */
#define MZ_ERR_DEVICE (-4)   /* a kernel flagged an internal inconsistency */

#define MZ_MAX_ACTIONS 64

/* How the per-selection tie-break uniforms of PUCT (mcts/action_selection.py:46) are supplied. */
#define MZ_NOISE_COUNTER 0   /* u = f(seed, tree id, call index, action); see csrc/noise.h */
#define MZ_NOISE_BUFFER 1    /* u = noise[tree][call][action] from a caller buffer */
#define MZ_NOISE_NONE 2      /* u = 0 */

/* Built-in leaf evaluators for the fused search (mz_search_run). */
#define MZ_NET_HASH 0        /* synthetic hash network (csrc/hashnet.h) */
#define MZ_NET_DENSE 1       /* network lowered to a dense program (mz_set_network), weights resident in LDS */
#define MZ_NET_MFMA 2        /* network in matrix-core layout (mz_set_network_mfma), n_actions <= 16 */

/* The attributes MCTS reads from its config (mcts/mcts.py:27-30,68-69,90,112,162,181-227,236). */
typedef struct mz_config {
    int32_t n_actions;              /* config.n_actions, 1..64 */
    int32_t n_simulations;          /* config.simulations */
    int32_t players;                /* config.players, 1 or 2 */
    int32_t gumbel;                 /* config.gumbel */
    int32_t hidden_size;            /* floats per hidden state (product of its dims) */
    int32_t max_batch;              /* trees the context is sized for */
    double discount;                /* config.gamma */
    double c_base, c_init;          /* PUCT */
    int32_t max_considered_actions; /* Gumbel */
    int32_t noise_mode;             /* MZ_NOISE_* (PUCT only) */
    double c_visit, c_scale;        /* Gumbel */
    uint64_t seed;                  /* MZ_NOISE_COUNTER key */
    const double *c_table_h;        /* optional host table [n_simulations+2] of
                                       c_init + log((n + c_base + 1)/c_base) as the caller's libm/NumPy
                                       evaluates it; NULL = computed with the C library's log() */
    int32_t device;                 /* HIP device ordinal */
    int32_t lanes_per_tree;         /* 0 = choose; else a power of two >= n_actions, <= 64 */
} mz_config;

typedef struct mz_ctx mz_ctx;

/* Host views of one tree, field for field the reference's Tree (mcts/tree.py:26-40).
 * All pointers are HOST buffers sized for N = n_simulations + 1 nodes. */
typedef struct mz_tree_host {
    int32_t *node_visits;        /* [N] */
    double *node_values;         /* [N] */
    double *node_raw_values;     /* [N] */
    int32_t *parents;            /* [N] */
    int32_t *action_from_parent; /* [N] */
    int32_t *to_play;            /* [N] */
    int32_t *children_index;     /* [N][A] */
    void *children_prior_logits; /* [N][A] double (PUCT) or float (Gumbel) */
    int32_t *children_visits;    /* [N][A] */
    double *children_rewards;    /* [N][A] */
    double *children_values;     /* [N][A] */
    float *hidden_states;        /* [N][H], may be NULL */
} mz_tree_host;

/* Replaces MCTS.__init__ (mcts/mcts.py:19-20) + the array allocation of
 * initialize_tree (mcts/mcts.py:68-92): allocates the tree storage for max_batch trees once. */
mz_ctx *mz_create(const mz_config *cfg);
void mz_destroy(mz_ctx *ctx);
const char *mz_last_error(void);

/* Key of the counter-based tie-break noise (MZ_NOISE_COUNTER) for the searches that follow: stands for the
 * position of the global np.random stream at the start of a reference search (mcts/action_selection.py:46 draws
 * from it), which a caller of the reference moves with np.random.seed / utils/utils.py:28-31 set_seed.
 * mz_create sets it from mz_config.seed; callers that search the same trees move after move pass a new key per
 * move so that no two searches replay the same uniforms. */
int mz_set_seed(mz_ctx *ctx, uint64_t seed);

/* Replaces add_exploration_noise (mcts/mcts.py:22-31) + mask_illegal_action_logits
 * (utils/game_utils.py:71-78) for B roots.  logits f32[B][A]; dirichlet f64[B][A] (PUCT) or NULL
 * (Gumbel); legal u8[B][A]; out_prior f64[B][A] (PUCT) / f32[B][A] (Gumbel).
 * shift_and_mask 0 stops after add_exploration_noise (PUCT; `legal` may then be NULL). */
int mz_root_prepare(mz_ctx *ctx, int32_t B, const float *logits, const double *dirichlet,
                    double exploration_frac, const uint8_t *legal, int32_t shift_and_mask,
                    void *out_prior, void *stream);

/* Replaces initialize_tree's root update (mcts/mcts.py:93-96, update_tree_node :37-57).
 * root_prior: prepared prior logits, f64[B][A] (PUCT) or f32[B][A] (Gumbel); root_value f64[B];
 * root_hidden f32[B][H] or NULL; root_to_play i32[B]; legal u8[B][A]; gumbel f64[B][A] (Gumbel only).
 * tree_id0: id of tree 0 for counter noise (a rank passes its first global tree index). */
int mz_tree_reset(mz_ctx *ctx, int32_t B, const void *root_prior, const double *root_value,
                  const float *root_hidden, const int32_t *root_to_play, const uint8_t *legal,
                  const double *gumbel, uint64_t tree_id0, void *stream);

/* Replaces MCTS.simulate (mcts/mcts.py:98-114) with the selection rules of
 * mcts/action_selection.py:25-49 (PUCT) / :73-118 (Gumbel root + interior) and the value
 * transforms of mcts/qtransforms.py, for every tree.  noise: f64[B][max_calls][A] when
 * noise_mode is MZ_NOISE_BUFFER, else NULL.  If out_parent_hidden (f32[B][H]) is not NULL it
 * receives hidden_states[parent], the input of recurrent_inference (mcts/mcts.py:241);
 * out_parent / out_action (i32[B], optional) receive the selected (node, action) pairs. */
int mz_select(mz_ctx *ctx, const double *noise, int32_t noise_max_calls, float *out_parent_hidden,
              int32_t *out_parent, int32_t *out_action, void *stream);

/* Device arrays written by the last mz_select: i32[B] each (parent node, action, next_to_play, depth). */
int mz_select_outputs(mz_ctx *ctx, const int32_t **parent, const int32_t **action,
                      const int32_t **next_to_play, const int32_t **depth);

/* Replaces MCTS.expand + MCTS.backward (mcts/mcts.py:116-164) for the leaf each tree selected.
 * logits f32[B][A], value f64[B], reward f64[B], next_hidden f32[B][H] (or NULL). */
int mz_expand_backup(mz_ctx *ctx, const float *logits, const double *value, const double *reward,
                     const float *next_hidden, void *stream);

/* Replaces muzero_action_selection_to_act / gumbel_muzero_action_selection_to_act
 * (mcts/action_selection.py:52-70, :121-138) and the return of MCTS.search (mcts/mcts.py:249-252).
 * u f64[B]: the uniform behind np.random.choice (PUCT; NULL = 0).  Outputs: action i32[B],
 * probs f64[B][A], root_value f64[B], q f64[B][A] (may be NULL). */
int mz_to_act(mz_ctx *ctx, double temperature, const double *u, int32_t *out_action,
              double *out_probs, double *out_root_value, double *out_q, void *stream);

/* Uploads a network lowered to a dense program: replaces the weights behind
 * network.recurrent_inference (network.py:105-126; MuZeroFullyConnectedNetwork :189-225, 405-479, or
 * MuZeroResidualNetwork :133-186 on a small board).  meta_h int32[n_meta] + data_h float[n_data] in the
 * level-scheduled layout documented in muzero_amd/packing.py; io_h int32[12] = scratch offsets
 * {hidden in, action in, next hidden, reward logits, policy logits, value logits} then {H, A,
 * action encoding size, support size, normalisation groups, observation size}. */
int mz_set_network(mz_ctx *ctx, const int32_t *meta_h, int32_t n_meta, const float *data_h, int32_t n_data,
                   const int32_t *io_h);

/* Uploads the same kind of network in matrix-core layout (muzero_amd/packing.py, pack_mfma) for the
 * batched leaf evaluation of mz_expand_backup_net: any lowered network whose 32-tree activation tile
 * fits in LDS, e.g. MuZeroResidualNetwork (network.py:133-186) on a 3x3 board.  steps_h int32[n_steps*12]
 * (one dense layer each, in n_levels dependency levels), sched_h int32[n_sched] = the host-made assignment of
 * every level's (step, column tile) jobs to the waves of a 32-tree and of a 16-tree block, wfrag_h float[n_w]
 * (the stored 16x16 weight blocks in v_mfma_f32_16x16x4_f32 operand order; all-zero blocks are left out),
 * bias_h float[n_b], ctinfo_h int32[n_jobs*3] = per (step, column tile) {weight offset in floats, first entry
 * in kidx, number of blocks}, kidx_h int32[n_kidx] = the 16-input block index of every stored block, stride =
 * activation floats per tree, plane_major != 0: the program keeps the hidden state as [cell][channel] with
 * io_h[10] channels (stored hidden states stay [channel][cell] as in the reference), io_h as in mz_set_network.
 * The table layouts are documented in muzero_amd/packing.py. */
int mz_set_network_mfma(mz_ctx *ctx, const int32_t *steps_h, int32_t n_steps, int32_t n_levels,
                        const int32_t *sched_h, int32_t n_sched, const float *wfrag_h, int64_t n_w,
                        const float *bias_h, int32_t n_b, const int32_t *ctinfo_h, int32_t n_jobs,
                        const int32_t *kidx_h, int32_t n_kidx, int32_t stride, int32_t plane_major,
                        const int32_t *io_h);

/* The two programs of network.initial_inference (network.py:86-103) in the same matrix-core layout, for
 * mz_initial_inference_mfma: part 0 = the representation network (observation -> raw hidden state; io_h[0] = where the
 * observation of io_h[11] floats goes in the activation row, io_h[2] = where the raw hidden state comes out), part 1 =
 * the prediction network on the scaled hidden state (io_h[0] = where the hidden state goes, io_h[8] = 0: no action
 * beside it, io_h[4] / io_h[5] = policy logits / value logits).  Other arguments and io_h entries as in
 * mz_set_network_mfma. */
int mz_set_root_network_mfma(mz_ctx *ctx, int32_t part, const int32_t *steps_h, int32_t n_steps, int32_t n_levels,
                             const int32_t *sched_h, int32_t n_sched, const float *wfrag_h, int64_t n_w,
                             const float *bias_h, int32_t n_b, const int32_t *ctinfo_h, int32_t n_jobs,
                             const int32_t *kidx_h, int32_t n_kidx, int32_t stride, int32_t plane_major,
                             const int32_t *io_h);

/* Replaces network.initial_inference (network.py:86-103, called by MCTS.search at mcts/mcts.py:176-180) for B
 * observations: representation network, per-group min-max scaling of the hidden state (utils/network_utils.py:132-158),
 * prediction network, support_to_scalar of the value (utils/network_utils.py:75-96), level by level on the matrix
 * cores.  obs float[B][obs_size], out_logits float[B][n_actions], out_hidden float[B][hidden_size] (the reference's
 * element order), out_value double[B]; all device pointers.  The outputs are what mz_root_prepare / mz_tree_reset take. */
int mz_initial_inference_mfma(mz_ctx *ctx, int32_t B, const float *obs, float *out_logits, float *out_hidden,
                              double *out_value, void *stream);

/* Replaces network.recurrent_inference (mcts/mcts.py:240-243) + MCTS.expand + MCTS.backward
 * (mcts/mcts.py:116-164) for the pending selection of every tree: leaves are evaluated on the matrix
 * cores by the network given to mz_set_network_mfma, then expanded and backed up.  With select_next != 0
 * (and simulations left) the same launch also runs the next MCTS.simulate (as mz_select without the
 * optional outputs; noise as in mz_select), so a search is mz_select once, then n_simulations calls. */
int mz_expand_backup_net(mz_ctx *ctx, int32_t select_next, const double *noise, int32_t noise_max_calls,
                         void *stream);

/* mz_expand_backup_net with every byte of its staging (the per-level activations of the batched leaf evaluation,
 * the leaves' logits / values / rewards handed to the expansion) in a fixed array inside the library's device image
 * instead of buffers obtained from hipMalloc.  No counterpart in the reference: it exists for callers that cannot
 * allocate device memory around the call - code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  The call itself performs no hipMalloc / hipFree and uses no allocating library routine.
 * Always evaluates level by level through that array (the path mz_expand_backup_net takes for batches below 4096
 * trees), so its results equal mz_expand_backup_net's on that path bit for bit.
 * Limit: rows * stride * 4 + B * (4 * n_actions + 16) bytes (rows = B rounded up to 32, stride = the activation floats
 * per tree given to mz_set_network_mfma, each part rounded up to 256 bytes) must not exceed 16 MiB; larger batches
 * are rejected with MZ_ERR_ARG.  Each GPU the library runs on holds the 16 MiB until the process ends.
 * All calls on one GPU share the array, whichever context they come from: calls that overlap in time on one GPU
 * (different streams or host threads) race; calls ordered on one stream do not. */
int mz_expand_backup_net_static(mz_ctx *ctx, int32_t select_next, const double *noise, int32_t noise_max_calls,
                                void *stream);

/* Uploads the representation and prediction programs behind network.initial_inference
 * (network.py:86-103).  io_h int32[8] = {obs offset in the representation scratch, raw hidden offset in it,
 * hidden offset in the prediction scratch, policy logits offset, value logits offset, observation
 * floats, support size, normalisation groups}. */
int mz_set_root_network(mz_ctx *ctx, const int32_t *rep_meta_h, int32_t rep_n_meta, const float *rep_data_h,
                        int32_t rep_n_data, const int32_t *pred_meta_h, int32_t pred_n_meta,
                        const float *pred_data_h, int32_t pred_n_data, const int32_t *io_h);

/* Replaces the head of MCTS.search (mcts/mcts.py:176-235) for B observations in one launch:
 * initial_inference, add_exploration_noise / Gumbel noise, mask_illegal_action_logits and
 * initialize_tree.  obs f32[B][obs floats]; legal u8[B][A]; to_play i32[B]; dirichlet f64[B][A] (PUCT)
 * or gumbel f64[B][A] (Gumbel); out_logits f32[B][A] / out_value f64[B] optional copies of the raw
 * network outputs. */
int mz_search_begin_net(mz_ctx *ctx, int32_t B, const float *obs, const uint8_t *legal,
                        const int32_t *to_play, const double *dirichlet, double exploration_frac,
                        const double *gumbel, uint64_t tree_id0, float *out_logits, double *out_value,
                        void *stream);

/* Fused search: all n_simulations of mcts/mcts.py:236-248 in one launch, leaves evaluated inside
 * the kernel.  net_kind MZ_NET_HASH: net_params_h = host int32[3] {logit_shift, reward_on,
 * parent_predicts}; MZ_NET_DENSE: the network given to mz_set_network (net_params_h unused), which
 * must fit in LDS together with the activation scratch.  Tree must have been reset. */
int mz_search_run(mz_ctx *ctx, int32_t net_kind, const void *net_params_h, const double *noise,
                  int32_t noise_max_calls, void *stream);

/* mz_search_run followed by mz_to_act (mcts/mcts.py:236-252 as one call): the simulations, then the final
 * action rule with the outputs of mz_to_act.  Where the search kernel can, the rule runs at the end of the
 * same launch (small batches, where a launch costs as much as the rule); otherwise two launches.  Same
 * results as the two calls. */
int mz_search_run_act(mz_ctx *ctx, int32_t net_kind, const void *net_params_h, const double *noise,
                      int32_t noise_max_calls, double temperature, const double *u, int32_t *out_action,
                      double *out_probs, double *out_root_value, double *out_q, void *stream);

/* One whole move of B searches as one call: mz_search_begin_net (roots from observations), mz_search_run with
 * MZ_NET_DENSE, mz_to_act - what MCTS.search (mcts/mcts.py:166-252) does for one position, for B positions.
 * Arguments as in those three calls.  Small batches of 2-action PUCT trees take two launches (roots; simulations
 * with the final rule), every other shape three.  Same results as the three calls. */
int mz_search_move(mz_ctx *ctx, int32_t B, const float *obs, const uint8_t *legal, const int32_t *to_play,
                   const double *dirichlet, double exploration_frac, const double *gumbel, uint64_t tree_id0,
                   const double *noise, int32_t noise_max_calls, double temperature, const double *u,
                   int32_t *out_action, double *out_probs, double *out_root_value, double *out_q, void *stream);

/* Copies tree b to host arrays (synchronises `stream`).  Nodes not yet expanded read as a fresh
 * tree's (mcts/mcts.py:72-92). */
int mz_tree_export(mz_ctx *ctx, int32_t b, mz_tree_host *out_h, void *stream);

/* Waits for `stream` and reports what kernels flagged since the last check: MZ_ERR_DEVICE when a selection state
 * was inconsistent, or when a search made more PUCT selection calls than the MZ_NOISE_BUFFER given to it holds
 * rows (the reference would have drawn further uniforms, mcts/action_selection.py:46; the engine used 0).  The
 * other entry points never synchronise; mz_tree_export and mz_node_backward run the same check. */
int mz_check(mz_ctx *ctx, void *stream);

/* Per-tree counters since the last reset: node_visits i64[B] = sum(node_visits) - 1
 * (device or NULL), noise_calls i32[B] (device or NULL). */
int mz_stats(mz_ctx *ctx, int64_t *out_node_visits, int32_t *out_noise_calls, void *stream);

/* Per-node entry points behind the reference's module-level functions.  They evaluate ONE rule at
 * node `node` of tree `b` of the context and copy the result to host.
 * which: 0 qtransform_by_parent_and_siblings (mcts/qtransforms.py:7-22)
 *        1 qtransform_completed_by_mix_value (mcts/qtransforms.py:25-48)
 *        2 compute_mixed_value vector        (mcts/qtransforms.py:51-77)
 * out_h f64[A]. */
int mz_node_qtransform(mz_ctx *ctx, int32_t b, int32_t node, int32_t which, double *out_h, void *stream);
/* Selection at one node: depth 0 = root rule, else interior (mcts/action_selection.py:11-22);
 * uniform_h f64[A] or NULL (PUCT tie-break draw).  Returns the action in *out_action_h. */
int mz_node_select(mz_ctx *ctx, int32_t b, int32_t node, int32_t depth, const double *uniform_h,
                   int32_t *out_action_h, void *stream);
/* MCTS.backward (mcts/mcts.py:143-164) alone, from node `leaf` of tree b with the leaf's to_play. */
int mz_node_backward(mz_ctx *ctx, int32_t b, int32_t leaf, int32_t to_play, void *stream);
/* Uploads host arrays as tree b (inverse of mz_tree_export); root_legal_h f64[A], gumbel_h f64[A] or NULL. */
int mz_tree_import(mz_ctx *ctx, int32_t b, const mz_tree_host *in_h, const double *root_legal_h,
                   const double *gumbel_h, void *stream);

/* Environments stepped on the device for batched self-play. */
#define MZ_ENV_TICTACTOE 1   /* games/tictactoe.py:9-81 (3x3), games/game.py:56-74 */
#define MZ_ENV_CARTPOLE 2    /* games/cartpole.py:9-48 = gymnasium CartPole-v1 (public specification) */

/* Caller-owned DEVICE state of B environments (one entry per tree of the context). */
typedef struct mz_env_state {
    int32_t kind;               /* MZ_ENV_* */
    int32_t max_moves;          /* config.max_moves: a game also ends when len(history) > max_moves */
    int32_t stale_observation;  /* 1 = search move k from the observation saved with move k-1, as
                                   self_play.py:86-91 + games/game.py:143-149 do; 0 = from the current one */
    int32_t obs_size;           /* floats per observation: 27 (TicTacToe planes) / 4 (CartPole) */
    void *state;                /* TicTacToe: int8[B][9] board (-1 X, +1 O, 0 empty); CartPole: f64[B][4] */
    int32_t *mover;             /* TicTacToe: [B] -1/+1 player about to move; CartPole: NULL */
    int32_t *to_play;           /* [B] game.to_play() for the next search (in/out) */
    uint8_t *legal;             /* [B][A] game.legal_actions() for the next search (in/out) */
    float *cur_obs;             /* [B][obs] current observation (in/out) */
    float *search_obs;          /* [B][n_stacked_observations * (1 + stack_action)][obs] what the next search starts
                                   from (in/out): one observation, or the stack described below */
    int32_t *moves;             /* [B] moves played in the current game (in/out) */
    const double *reset_state;  /* CartPole: f64[B][4] initial states, uniform(-0.05, 0.05), consumed when an
                                   environment restarts; TicTacToe: NULL */
    int32_t n_stacked_observations; /* config.n_stacked_observations; 0 counts as 1 */
    int32_t stack_action;       /* config.stack_action: 1 = every observation of the stack is followed by a plane
                                   filled with action / n_actions.  With more than one observation or with action
                                   planes, search_obs is GameHistory.stack_n_observations(-1, ...) of the game so far
                                   (games/game.py:120-160, called at self_play.py:86-91): newest saved observation
                                   first, zero planes where the game is younger than the stack, and for a game that
                                   has not started the initial observation followed by zero planes.  That rule is the
                                   stale one by construction, so stale_observation must be 1. */
} mz_env_state;

/* Replaces one iteration of SelfPlay.play's loop after the search (self_play.py:99-113) for every
 * environment: the final action rule (as mz_to_act), game.step(action), the GameHistory.save record of
 * this move (out_obs_before f32[B][obs] may be NULL, out_action, out_reward f32[B], out_to_play i32[B],
 * out_probs, out_root_value), termination (out_done u8[B]: terminated or len(history) > max_moves) and
 * the restart of finished games.  env_h is a HOST struct of device pointers. */
int mz_act_env_step(mz_ctx *ctx, const mz_env_state *env_h, double temperature, const double *u,
                    int32_t *out_action, double *out_probs, double *out_root_value, float *out_obs_before,
                    float *out_reward, uint8_t *out_done, int32_t *out_to_play, void *stream);

/* The moves of B environments kept on the device in the order they were played: row k holds move k of the current
 * collection window for every environment ([capacity][n_envs] per field), i.e. what SelfPlay.play appends to its
 * GameHistory per move (self_play.py:107-109 -> games/game.py:99-112) plus the end-of-game flag.  Caller-owned
 * DEVICE arrays in a HOST struct; written by mz_act_env_step_log, read by mz_log_count_games / mz_log_list_games
 * and, through mz_trajectories, by mz_make_targets - the records never visit the host. */
typedef struct mz_move_log {
    int32_t capacity, n_envs, obs_size, n_actions;
    float *observations;   /* [capacity][n_envs][obs_size] the observation stored with the move */
    int32_t *actions;      /* [capacity][n_envs] */
    double *rewards;       /* [capacity][n_envs] */
    int32_t *to_plays;     /* [capacity][n_envs] */
    double *root_values;   /* [capacity][n_envs] */
    double *probs;         /* [capacity][n_envs][n_actions] */
    uint8_t *done;         /* [capacity][n_envs] 1 = the game ended with this move */
    const uint8_t *open_at_start; /* [n_envs] or NULL. 1 = the environment was in the middle of a game when row 0 was
                                   * written: its moves up to the first end-of-game flag are the tail of a game that began
                                   * before this window and are not listed as a game (SelfPlay.play stores whole games
                                   * only, self_play.py:77-115).  NULL = every environment started a game at row 0. */
} mz_move_log;

/* mz_act_env_step that also appends every environment's record to row `row` of `log_h` (0 <= row < capacity).
 * The out_* arrays may each be NULL when the caller only wants the log. */
int mz_act_env_step_log(mz_ctx *ctx, const mz_env_state *env_h, const mz_move_log *log_h, int32_t row,
                        double temperature, const double *u, int32_t *out_action, double *out_probs,
                        double *out_root_value, float *out_obs_before, float *out_reward, uint8_t *out_done,
                        int32_t *out_to_play, void *stream);

/* The finished games inside rows [0, n_rows) of a log, environment by environment and in playing order inside an
 * environment (the batched counterpart of the game list a replay buffer accumulates, replay_buffer.py:30-57).
 * mz_log_count_games: out_count i32[n_envs] = games of each environment that ended inside the rows (moves after an
 * environment's last finished game belong to a game still open and are not listed; so are, with open_at_start, the
 * moves before the first end of a game that was already running when the window opened).
 * mz_log_list_games: first_game i64[n_envs] = exclusive prefix sum of those counts (the caller's scan);
 * out_base i64[G] = index of a game's first position, out_len i32[G] = its length; position i of game g is at
 * index out_base[g] + i * n_envs of the log's per-move arrays (see mz_trajectories.game_base). */
int mz_log_count_games(const mz_move_log *log_h, int32_t n_rows, int32_t *out_count, void *stream);
int mz_log_list_games(const mz_move_log *log_h, int32_t n_rows, const int64_t *first_game, int64_t *out_base,
                      int32_t *out_len, void *stream);

/* Finished games kept on the device: the fields of GameHistory (games/game.py:83-94) as per-position arrays.
 * Flat form (game_base NULL): positions of game g at [game_offset[g], game_offset[g+1]).
 * Strided form (game_base given, e.g. over a mz_move_log): position i of game g at game_base[g] + i *
 * position_stride; game_offset is still the running sum of the game lengths.  HOST struct of DEVICE pointers. */
typedef struct mz_trajectories {
    int32_t n_games, obs_size, n_actions, reserved;
    const int64_t *game_offset;     /* [n_games + 1] */
    const float *observations;      /* [T][obs_size] */
    const int32_t *actions;         /* [T] */
    const double *rewards;          /* [T] */
    const int32_t *to_plays;        /* [T] */
    const double *root_values;      /* [T] */
    const double *probs;            /* [T][n_actions] action_probabilities */
    const double *re_root_values;   /* [T] reanalysed_root_values, or NULL */
    const double *re_probs;         /* [T][n_actions] reanalysed_action_probabilities, or NULL */
    const uint8_t *has_reanalysed;  /* [n_games] 1 where save_reanalysed_stats was applied, or NULL */
    const int64_t *game_base;       /* [n_games] or NULL */
    int64_t position_stride;        /* used with game_base */
} mz_trajectories;

/* Replaces the per-game body of ReplayBuffer.sample (replay_buffer.py:70-101) for n_samples sampled
 * positions (game_idx[i], t[i]): GameHistory.stack_n_observations (games/game.py:120-160) ->
 * out_obs f32[n][n_stack*(1+stack_action)*obs], the unrolled one-hot actions with absorbing padding ->
 * out_actions f32[n][U+1][A] (absorbed i32[n][U+1]: the sampled earlier positions used past the end of
 * a game, left-aligned), GameHistory.make_target (games/game.py:185-253) -> out_value, out_reward
 * f32[n][U+1], out_policy f32[n][U+1][A].  gamma_pow: DEVICE f64[td_steps+1] = gamma**k as the host
 * evaluates it.  All other pointers are device pointers; the sampling itself stays with the caller. */
int mz_make_targets(const mz_trajectories *traj_h, int32_t n_samples, const int32_t *game_idx, const int32_t *t,
                    const int32_t *absorbed, int32_t td_steps, int32_t unroll_steps, double gamma,
                    const double *gamma_pow, int32_t n_stack, int32_t stack_action, float *out_obs,
                    float *out_actions, float *out_value, float *out_reward, float *out_policy, void *stream);

/* scalar_to_support with the default value transform (utils/network_utils.py:70-71, 99-129), which the
 * trainer applies to the value and reward targets (trainer.py:70-71): x f32[n] -> out f32[n][2L+1].
 * Device pointers; needs no context. */
int mz_scalar_to_support(const float *x, int64_t n, int32_t support_limit, float *out, void *stream);

/* Sequential-halving schedule, mcts/action_selection.py:160-175 (host, pure integer). out_h i32[S]. */
int mz_considered_visits(int32_t max_considered_actions, int32_t simulations, int32_t *out_h);

/* Root noise of B consecutive PUCT searches out of NumPy's legacy generator (host, no GPU): per search one
 * np.random.dirichlet([alpha] * A) row (mcts/mcts.py:27) and - with_uniform != 0 - the np.random.random_sample() that
 * np.random.choice draws for the final action (mcts/action_selection.py:69), in that order, search after search.
 * mt_key u32[624] and *mt_pos are the MT19937 state of np.random.get_state(), advanced in place to where B such searches
 * leave the stream.  alpha must lie in (0, 1) (the rejection sampler of NumPy's legacy standard_gamma for shapes below
 * one is restated; other shapes: MZ_ERR_ARG, draw with NumPy).  out_dirichlet f64[B][A], out_u f64[B] or NULL. */
int mz_legacy_root_noise(uint32_t *mt_key, int32_t *mt_pos, double alpha, int32_t n_actions, int32_t B,
                         int32_t with_uniform, double *out_dirichlet, double *out_u);

/* np.random.gumbel(size=n) of the same legacy generator - the draw of gumbel_noise in MCTS.search (mcts/mcts.py:211),
 * for n = B * n_actions the draws of B consecutive searches: out f64[n] = -log(-log(1 - next_double())) in stream
 * order; state in and out as for mz_legacy_root_noise.  Host only; the logarithms run on a few host threads. */
int mz_legacy_gumbel(uint32_t *mt_key, int32_t *mt_pos, int64_t n, double *out);

/* Library/ABI version. */
int mz_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MUZERO_AMD_H */
