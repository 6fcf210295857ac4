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
// k_search_gcache<tm, A, row width>: the whole simulation loop over cached decisions (Gumbel, 2..16 actions, records
// with 16- or 32-bit visit / index rows).  `supported` tells whether the tree layout is one the kernels are built for
// and everything fits in LDS with tm-tree tiles (tm = 32 or 16).
bool search_gcache_supported(const Dev &d, const MfmaNet &net, int tm);
hipError_t launch_search_gcache(int tm, hipStream_t st, const Dev &d, const MfmaNet &net, int first_leaf, int n_sims);
// Level by level through global memory: k_stage_inputs, k_net_level_rows (one level for a range of 32-tree tiles),
// and for whole searches k_leaf_backup_select.
#define MZ_DEC_LDS 256  // nodes of a tree whose cached PUCT decisions k_leaf_backup_select keeps in LDS (larger trees: in memory)
#define MZ_LVL_DESC 40  // ints per block descriptor of k_net_level_rows: KK, in_off, N, out_off, skip_off, act, bias_off,
                        // n column tiles, first column tile, dense flag, then per wave: weight byte offset, blocks, kidx offset
static_assert(MZ_LVL_DESC >= 10 + 3 * MZ_LVL_WAVES, "a block descriptor holds three ints per wave after its ten header ints");
#define MZ_LVL_AHEAD 3 // ... of which it requests this many before its barrier, the others one by one behind the blocks that multiply
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

// The whole search in one launch, phases chained through tagged data in global memory (search_chain.h).
#define MZ_CH_TM 32      // trees per tile
#define MZ_CH_NT 512     // threads per workgroup: waves 0-3 network, 4-7 trees
#define MZ_CH_SEG 40     // ints per segment descriptor
#define MZ_CH_RING 6     // weight blocks (16 inputs each) of a column tile in flight or in use
#define MZ_CH_MAXB 24    // weight blocks of a column tile: layers of up to 384 inputs
#define MZ_CH_NF 18      // granule pairs of a segment's inputs that a thread fetches ahead (inputs of up to 288 floats)
#define MZ_CH_MAXL 16    // network levels
#define MZ_CH_MAXSEG 16  // segments of one workgroup over all levels (kept in LDS)
#define MZ_CH_PW 36      // floats per row of a wave's output patch (two column tiles + padding)

// Segment descriptor: 0 K (floats, multiple of 16), 1 input offset, 2 output offset of the first column tile,
// 3 residual offset of the first column tile or -1, 4 activation, 5 first bias entry, 6 column tiles (1..8),
// 7 dense (every tile multiplies input blocks 0, 1, 2, ...), 8 producing phase of the input (0: the tree waves,
// l + 1: network level l), 9 producing phase of the residual input, 10 own phase (level + 1), 11 unused,
// then per column tile: weight byte offset, blocks, first entry in kidx.
struct ChainArgs {
    const int32_t *seg;        // [segments][MZ_CH_SEG]
    const int32_t *blk_range;  // [levels][G + 1]: segments of workgroup m of a cluster at a level
    float *acts2;              // [rows][stride] granules {float value, uint32 tag}
    unsigned acts2_bytes;
    int G, n_clusters, n_tiles;
    int kk_max;                // widest segment input (floats)
    int in_lo, in_n;           // the network inputs inside a row: first float, floats (multiple of 16)
    int out_phase[4];          // producing phase of next state, reward logits, policy logits, value logits
    int first_leaf, n_sims;
    const double *noise;
    int noise_max_calls;
    long long timeout;         // ticks of the 100 MHz clock that one wait may last
};

size_t search_chain_lds(const MfmaNet &net, int kk_max, bool gumbel);
// rl: action lanes of the tree waves (instantiated: see chain_kernels.hip); false: not built for this shape
bool search_chain_built(int rl);
hipError_t launch_search_chain(int rl, bool gumbel, hipStream_t st, const Dev &d, const MfmaNet &net, const ChainArgs &ca,
                               unsigned n_blocks);


}  // namespace mz
