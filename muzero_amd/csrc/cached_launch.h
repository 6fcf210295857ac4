// Launch functions of the cached-decision search (search_cached.h), one kernel per (tile size, action count, row
// width).  The instantiations are spread over three translation units (cached_kernels_[abc].hip: 2..6, 7..11 and 12..16
// actions) so that they compile side by side.
#pragma once
#include "search_cached.h"

namespace mz {

// mz_create gives visit / index rows 16-bit entries only where that shortens the 64-byte-granular record.
template <int A>
constexpr bool gumbel_rows16_exist() { return GumbelLayout<A, 2>::rec < GumbelLayout<A, 4>::rec; }

template <int A, int IW>
hipError_t launch_gcache_kernel(int tm, hipStream_t st, const Dev &d, const MfmaNet &net, int first_leaf, int n_sims, size_t lds)
{
    auto kern = tm == 32 ? k_search_gcache<32, A, IW> : k_search_gcache<16, A, IW>;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((d.B + tm - 1) / tm), dim3(tm * 16), lds, st, d, net, first_leaf, n_sims);
    return hipSuccess;
}

template <int A>
hipError_t launch_gcache_actions(int tm, hipStream_t st, const Dev &d, const MfmaNet &net, int first_leaf, int n_sims, size_t lds)
{
    if (!d.L.idx16) return launch_gcache_kernel<A, 4>(tm, st, d, net, first_leaf, n_sims, lds);
    if constexpr (gumbel_rows16_exist<A>()) return launch_gcache_kernel<A, 2>(tm, st, d, net, first_leaf, n_sims, lds);
    return hipErrorInvalidValue;   // no context has such records
}

}  // namespace mz
