// The chained whole-search kernel (search_chain.h) and its launch function, a translation unit of its own.
#include "search_chain.h"

namespace mz {

size_t search_chain_lds(const MfmaNet &net, int kk_max, bool gumbel)
{
    return ((size_t)2 * MZ_CH_TM * (kk_max + 4) + 4 * MZ_CH_TM * MZ_CH_PW + (size_t)4 * net.stride) * 4 + 4 * 64 * 8 +
           (gumbel ? 0 : (size_t)MZ_CT_LDS * 8) + 64 + (MZ_CH_MAXL + 4) * 4 + (size_t)MZ_CH_MAXSEG * MZ_CH_SEG * 4;
}

bool search_chain_built(int rl) { return rl == 32; }

hipError_t launch_search_chain(int rl, bool gumbel, hipStream_t st, const Dev &d, const MfmaNet &net, const ChainArgs &ca,
                               unsigned n_blocks)
{
    const size_t lds = search_chain_lds(net, ca.kk_max, gumbel);
    hipError_t err = hipErrorInvalidValue;
    auto launch = [&](auto kern) {
        err = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
        if (err == hipSuccess) hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(MZ_CH_NT), lds, st, d, net, ca);
    };
    if (rl == 32) { if (gumbel) launch(k_search_chain<32, true>); else launch(k_search_chain<32, false>); }
    return err;
}

}  // namespace mz

#ifdef MZ_STAMPS
// Diagnostic build only: the per-workgroup phase stamps of the last chained search.
extern "C" int mz_debug_read_chain_stamps(long long *out_h, int n)
{
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return hipMemcpyFromSymbol(out_h, HIP_SYMBOL(mz::mz_chain_stamps), (size_t)n * 8) == hipSuccess ? 0 : -2;
}
#endif
