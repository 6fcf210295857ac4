// Cached-decision search kernels for 12..16 actions (see cached_launch.h).
#include "cached_launch.h"

namespace mz {

hipError_t launch_search_gcache_c(int tm, hipStream_t st, const Dev &d, const MfmaNet &net, int first_leaf, int n_sims, size_t lds)
{
    switch (d.L.A) {
    case 12: return launch_gcache_actions<12>(tm, st, d, net, first_leaf, n_sims, lds);
    case 13: return launch_gcache_actions<13>(tm, st, d, net, first_leaf, n_sims, lds);
    case 14: return launch_gcache_actions<14>(tm, st, d, net, first_leaf, n_sims, lds);
    case 15: return launch_gcache_actions<15>(tm, st, d, net, first_leaf, n_sims, lds);
    case 16: return launch_gcache_actions<16>(tm, st, d, net, first_leaf, n_sims, lds);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mz
