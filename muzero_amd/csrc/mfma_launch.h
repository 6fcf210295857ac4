// What the C ABI layer (engine.hip) needs of the matrix-core kernels, which are compiled in a translation unit of
// their own (mfma_kernels.hip): the network descriptor and one launch function per kernel family.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_common.h"

#define MZ_DESC_INTS 13  // ints per scheduled job (muzero_amd/packing.py DESC_INTS)
#define MZ_LVL_WAVES 8  // column tiles (= waves) per block of the level kernel (k_net_level_rows)

namespace mz {

struct MfmaNet {
    const int32_t *steps;        // 12 ints per step
    const int32_t *wstart8, *wdesc8;  // per-wave schedule of a 32-tree tile (8 waves): job ranges per level and
                                      // MZ_DESC_INTS ints per job, see muzero_amd/packing.py
    const int32_t *wstart4, *wdesc4;  // ... of a 16-tree tile (4 waves)
    int sched_ok8, sched_ok4;         // 0: the network has no schedule for that tile size (more than 64 jobs for a
                                      // wave): only the level-by-level kernels serve it
    int w_bytes, b_bytes;             // sizes of wfrag and bias in bytes (bounds of the buffer loads)
    const float *wfrag;
    const float *bias;
    const int32_t *ctinfo;       // per (step, column tile) job: weight offset (floats), first entry in kidx, number of blocks
    const int32_t *kidx;         // 16-input block indices of the jobs' stored blocks
    int n_steps, n_levels, stride;
    int plane_major;             // the hidden state inside the activation row is [cell][channel] (channels = io[10])
    int io[12];  // in_hidden, in_action, out_next, out_reward, out_policy, out_value, H, A, action_size, support, norm_groups, obs
};

// Raises a kernel's dynamic-LDS limit on the current device when `bytes` exceeds what was granted before (engine.hip).
hipError_t ensure_dynamic_lds(const void *kern, size_t bytes);

// Each returns the error of raising the LDS limit, if any; launch errors are read with hipGetLastError by the caller.
// k_net_step<tm>: one simulation's leaves, activations of a tm-tree tile in LDS.
hipError_t launch_net_step(int tm, hipStream_t st, const Dev &d, const MfmaNet &net, int leaf, float *logits, double *value,
                           double *reward);
// k_net_epilogue<lanes>: lanes = 64 or 16 per tree.
void launch_net_epilogue(int lanes, hipStream_t st, const Dev &d, const MfmaNet &net, int leaf, const float *acts,
                         float *logits, double *value, double *reward);
// k_search_mfma<32, rl, gumbel>: the whole simulation loop, every node of the path scored on the way down.
hipError_t launch_search_mfma(int rl, bool gumbel, hipStream_t st, const Dev &d, const MfmaNet &net, int first_leaf,
                              int n_sims, const double *noise, int noise_max_calls);
// k_search_gcache<tm, 9>: the whole simulation loop over cached decisions (Gumbel).  `supported` tells whether the
// kernel is built for this tree layout and fits in LDS with tm-tree tiles.
bool search_gcache_supported(const Dev &d, const MfmaNet &net, int tm);
hipError_t launch_search_gcache(int tm, hipStream_t st, const Dev &d, const MfmaNet &net, int first_leaf, int n_sims);
// Level by level through global memory: k_stage_inputs, k_net_level_rows (one level for a range of 32-tree tiles),
// and for whole searches k_leaf_backup_select.
#define MZ_LVL_DESC 40  // ints per block descriptor of k_net_level_rows: KK, in_off, N, out_off, skip_off, act, bias_off,
                        // n column tiles, first column tile, dense flag, then per wave: weight byte offset, blocks, kidx offset
static_assert(MZ_LVL_DESC >= 10 + 3 * MZ_LVL_WAVES, "a block descriptor holds three ints per wave after its ten header ints");
#define MZ_LVL_KB 20   // weight blocks of a column tile that k_net_level_rows requests before it copies its inputs
hipError_t launch_net_level_rows(unsigned n_jobs, unsigned n_row_tiles, int max_k16, hipStream_t st, const MfmaNet &net,
                                 const int32_t *desc, float *acts, int row_first, int level);
void launch_stage_inputs(hipStream_t st, const Dev &d, const MfmaNet &net, float *acts);
// initial inference on the level kernels: observations in, scaled hidden states across, logits and values out
void launch_root_stage_obs(hipStream_t st, int B, const MfmaNet &rep, const float *obs, float *acts);
void launch_root_hidden(hipStream_t st, int B, const MfmaNet &rep, const float *acts, float *out_hidden);
void launch_root_stage_hidden(hipStream_t st, int B, const MfmaNet &pred, const float *hidden, float *acts);
void launch_root_outputs(hipStream_t st, int B, const MfmaNet &pred, const float *acts, float *out_logits, double *out_value);
bool launch_leaf_backup_select(int rl, bool gumbel, hipStream_t st, const Dev &d, const MfmaNet &net, int t_first, int t_end,
                               int leaf, float *acts, const double *noise, int noise_max_calls, int select_next);

}  // namespace mz
