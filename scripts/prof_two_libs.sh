#!/bin/bash
# Per-kernel durations of one workload under two builds of the library: scripts/prof_two_libs.sh <workload> <lib A> <lib B>
cd "$(dirname "$0")/.."
root=$PWD; wl=$1; shift
export TMPDIR=/tmp MZ_BENCH_MIN_S=0.5
for lib in "$@"; do
  export MUZERO_AMD_LIB=$root/$lib
  tag=$(basename $lib .so)
  (cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/two_$tag -o r --output-format csv -- \
     python3 $root/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $root/gpurun_out/two_$tag.log 2>&1) || { echo failed $lib; exit 1; }
  echo "== $lib"
  head -4 gpurun_out/two_$tag/r_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,100-200
done
