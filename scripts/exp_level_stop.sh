#!/bin/bash
# GPU box, diagnostic build: duration of the level launches of the Atari-shaped search when the level kernel ends at
# once (1), behind its barrier (2), before its output stores (3) or runs whole (0) - what a launch costs apart from its work.
root=${GRAFT_REPO_ROOT:-$(pwd)}
export MUZERO_AMD_LIB=$root/muzero_amd/libmuzero_amd_stamps.so
for stop in 0 1 2 3; do
    echo "MZ_LVL_STOP=$stop"
    MZ_LVL_STOP=$stop bash $root/scripts/prof_by_grid.sh cfg5 2 | grep -E "k_net_level_rows.*x (13|8) |k_leaf_backup" || exit 1
done
