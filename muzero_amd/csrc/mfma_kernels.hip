// The matrix-core kernels (net_mfma.h) and their launch functions, compiled apart from the C ABI layer; the cached-decision
// search kernels (search_cached.h) are instantiated in cached_kernels_[abc].hip and dispatched from here.
#include <algorithm>
#include <cstdlib>
#include "search_cached.h"  // net_mfma.h, dense_net.h, tree_kernels.h

namespace mz {

hipError_t launch_net_step(int tm, hipStream_t st, const Dev &d, const MfmaNet &net, int leaf, float *logits, double *value,
                           double *reward)
{
    const size_t lds = (size_t)net.stride * tm * 4;
    auto kern = tm == 32 ? k_net_step<32> : k_net_step<16>;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((d.B + tm - 1) / tm), dim3(tm * 16), lds, st, d, net, leaf, logits, value, reward);
    return hipSuccess;
}

void launch_net_epilogue(int lanes, hipStream_t st, const Dev &d, const MfmaNet &net, int leaf, const float *acts,
                         float *logits, double *value, double *reward)
{
    if (lanes == 64)
        hipLaunchKernelGGL(k_net_epilogue<64>, dim3((d.B + 3) / 4), dim3(256), 0, st, d, net, leaf, acts, logits, value, reward);
    else
        hipLaunchKernelGGL(k_net_epilogue<16>, dim3((d.B + 15) / 16), dim3(256), 0, st, d, net, leaf, acts, logits, value, reward);
}

hipError_t launch_search_mfma(int rl, bool gumbel, hipStream_t st, const Dev &d, const MfmaNet &net, int first_leaf,
                              int n_sims, const double *noise, int noise_max_calls)
{
    constexpr int tm = 32;
    const size_t lds = (size_t)net.stride * tm * 4;
    hipError_t err = hipSuccess;
    auto launch = [&](auto kern) {
        err = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
        if (err == hipSuccess)
            hipLaunchKernelGGL(kern, dim3((d.B + tm - 1) / tm), dim3(tm * 16), lds, st, d, net, first_leaf, n_sims, noise,
                               noise_max_calls);
    };
    switch (rl) {
    case 2: if (gumbel) launch(k_search_mfma<tm, 2, true>); else launch(k_search_mfma<tm, 2, false>); break;
    case 4: if (gumbel) launch(k_search_mfma<tm, 4, true>); else launch(k_search_mfma<tm, 4, false>); break;
    case 8: if (gumbel) launch(k_search_mfma<tm, 8, true>); else launch(k_search_mfma<tm, 8, false>); break;
    default: if (gumbel) launch(k_search_mfma<tm, 16, true>); else launch(k_search_mfma<tm, 16, false>); break;
    }
    return err;
}

hipError_t launch_net_level_rows(unsigned n_jobs, unsigned n_row_tiles, int max_k16, hipStream_t st, const MfmaNet &net,
                                 const int32_t *desc, float *acts, int row_first, int level)
{
    static_assert(MZ_LVL_KB == 2 * MZ_WPF, "k_net_level_rows holds one or two rounds of weight blocks");
    const size_t lds = (size_t)(max_k16 * 16 + 4) * 32 * 4;
    auto kern = max_k16 <= MZ_WPF ? k_net_level_rows<MZ_WPF> : k_net_level_rows<2 * MZ_WPF>;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
#ifdef MZ_STAMPS
    if (const char *e = getenv("MZ_LVL_STOP")) level |= atoi(e) << 8;
#endif
    hipLaunchKernelGGL(kern, dim3(n_row_tiles, n_jobs), dim3(MZ_LVL_WAVES * 64), lds, st, net, desc, acts, row_first, level);
    return hipSuccess;
}

void launch_stage_inputs(hipStream_t st, const Dev &d, const MfmaNet &net, float *acts)
{
    hipLaunchKernelGGL(k_stage_inputs, dim3((d.B + 3) / 4), dim3(256), 0, st, d, net, acts);
}

void launch_root_stage_obs(hipStream_t st, int B, const MfmaNet &rep, const float *obs, float *acts)
{
    hipLaunchKernelGGL(k_root_stage_obs, dim3((B + 3) / 4), dim3(256), 0, st, B, rep, obs, acts);
}

void launch_root_hidden(hipStream_t st, int B, const MfmaNet &rep, const float *acts, float *out_hidden)
{
    if (rep.io[10] >= 8) hipLaunchKernelGGL(k_root_hidden<16>, dim3((B + 15) / 16), dim3(256), 0, st, B, rep, acts, out_hidden);
    else hipLaunchKernelGGL(k_root_hidden<64>, dim3((B + 3) / 4), dim3(256), 0, st, B, rep, acts, out_hidden);
}

void launch_root_stage_hidden(hipStream_t st, int B, const MfmaNet &pred, const float *hidden, float *acts)
{
    hipLaunchKernelGGL(k_root_stage_hidden, dim3((B + 3) / 4), dim3(256), 0, st, B, pred, hidden, acts);
}

void launch_root_outputs(hipStream_t st, int B, const MfmaNet &pred, const float *acts, float *out_logits, double *out_value)
{
    hipLaunchKernelGGL(k_root_outputs, dim3((B + 3) / 4), dim3(256), 0, st, B, pred, acts, out_logits, out_value);
}

bool launch_leaf_backup_select(int rl, bool gumbel, hipStream_t st, const Dev &d, const MfmaNet &net, int t_first, int t_end,
                               int leaf, float *acts, const double *noise, int noise_max_calls, int select_next)
{
    const unsigned grid = (unsigned)(((long long)(t_end - t_first) * 64 + MZ_BLOCK - 1) / MZ_BLOCK);
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(MZ_BLOCK), 0, st, d, net, t_first, t_end, leaf, acts, noise, noise_max_calls,
                           select_next);
    };
#define MZ_LBS(R) case R: if (gumbel) launch(k_leaf_backup_select<R, true>); else launch(k_leaf_backup_select<R, false>); return true;
    switch (rl) {
    MZ_LBS(2) MZ_LBS(4) MZ_LBS(8) MZ_LBS(16) MZ_LBS(32) MZ_LBS(64)
    default: return false;
    }
#undef MZ_LBS
}

static size_t gcache_dynamic_lds(const Dev &d, const MfmaNet &net, int tm)
{
    // activations (or the record staging of the re-decision phase, one area per wave, where that is larger), decisions,
    // paths, work list
    const size_t act = std::max((size_t)net.stride * tm * 4, (size_t)(tm / 4) * MZ_GC_STAGE);
    return act + ((size_t)d.L.N + 2 * d.L.S + 1) * tm * 4;
}

bool search_gcache_supported(const Dev &d, const MfmaNet &net, int tm)
{
    const Layout &L = d.L;
    // the record the kernels are instantiated for: GumbelLayout<A, 2 or 4> of search_cached.h, here from the run-time A
    auto ru = [](int x, int m) { return (x + m - 1) / m * m; };
    const int A = L.A, iw = L.idx16 ? 2 : 4;
    const int off_crew = 32 + 8 * A, off_prior = off_crew + 8 * A, off_cvis = off_prior + ru(4 * A, 8);
    const int off_cidx = off_cvis + ru(iw * A, 4), rec = ru(off_cidx + iw * A, 64);
    const bool layout = L.prior_f32 && A >= 2 && A <= 16 && L.off_cval == 32 && L.rec == rec && L.off_crew == off_crew &&
                        L.off_prior == off_prior && L.off_cvis == off_cvis && L.off_cidx == off_cidx;
    const size_t stat = (size_t)tm * 16 * 8 + (size_t)tm * A * 8 + (size_t)tm * 20 + 64;  // the kernel's static arrays
    const bool sched = tm == 32 ? net.sched_ok8 : net.sched_ok4;
    // the re-decision phase addresses a tile's records by 32-bit offsets; a work-list entry holds a node number of 22 bits
    const bool stage = (long long)tm * L.tree_stride < 0x7fffffffLL && L.N < (1 << 22);
    return layout && sched && stage && (tm == 32 || tm == 16) && gcache_dynamic_lds(d, net, tm) + stat <= 160 * 1024;
}

hipError_t launch_search_gcache_a(int tm, hipStream_t st, const Dev &d, const MfmaNet &net, int first_leaf, int n_sims, size_t lds);
hipError_t launch_search_gcache_b(int tm, hipStream_t st, const Dev &d, const MfmaNet &net, int first_leaf, int n_sims, size_t lds);
hipError_t launch_search_gcache_c(int tm, hipStream_t st, const Dev &d, const MfmaNet &net, int first_leaf, int n_sims, size_t lds);

hipError_t launch_search_gcache(int tm, hipStream_t st, const Dev &d, const MfmaNet &net, int first_leaf, int n_sims)
{
    const size_t lds = gcache_dynamic_lds(d, net, tm);
    if (d.L.A <= 6) return launch_search_gcache_a(tm, st, d, net, first_leaf, n_sims, lds);
    if (d.L.A <= 11) return launch_search_gcache_b(tm, st, d, net, first_leaf, n_sims, lds);
    return launch_search_gcache_c(tm, st, d, net, first_leaf, n_sims, lds);
}

}  // namespace mz

#ifdef MZ_STAMPS
// Diagnostic build only: the per-block phase stamps of the last k_net_step launch.
extern "C" int mz_debug_read_net_stamps(long long *out_h, int n)
{
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return hipMemcpyFromSymbol(out_h, HIP_SYMBOL(mz::mz_dbg_stamps), (size_t)n * 8) == hipSuccess ? 0 : -2;
}
#endif
