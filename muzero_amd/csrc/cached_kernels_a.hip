// Cached-decision search kernels for 2..6 actions (see cached_launch.h).
#include "cached_launch.h"

namespace mz {

hipError_t launch_search_gcache_a(int tm, hipStream_t st, const Dev &d, const MfmaNet &net, int first_leaf, int n_sims, size_t lds)
{
    switch (d.L.A) {
    case 2: return launch_gcache_actions<2>(tm, st, d, net, first_leaf, n_sims, lds);
    case 3: return launch_gcache_actions<3>(tm, st, d, net, first_leaf, n_sims, lds);
    case 4: return launch_gcache_actions<4>(tm, st, d, net, first_leaf, n_sims, lds);
    case 5: return launch_gcache_actions<5>(tm, st, d, net, first_leaf, n_sims, lds);
    case 6: return launch_gcache_actions<6>(tm, st, d, net, first_leaf, n_sims, lds);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mz
