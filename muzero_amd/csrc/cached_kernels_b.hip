// Cached-decision search kernels for 7..11 actions (see cached_launch.h).
#include "cached_launch.h"

namespace mz {

hipError_t launch_search_gcache_b(int tm, hipStream_t st, const Dev &d, const MfmaNet &net, int first_leaf, int n_sims, size_t lds)
{
    switch (d.L.A) {
    case 7: return launch_gcache_actions<7>(tm, st, d, net, first_leaf, n_sims, lds);
    case 8: return launch_gcache_actions<8>(tm, st, d, net, first_leaf, n_sims, lds);
    case 9: return launch_gcache_actions<9>(tm, st, d, net, first_leaf, n_sims, lds);
    case 10: return launch_gcache_actions<10>(tm, st, d, net, first_leaf, n_sims, lds);
    case 11: return launch_gcache_actions<11>(tm, st, d, net, first_leaf, n_sims, lds);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mz
