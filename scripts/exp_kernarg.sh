cd $GRAFT_REPO_ROOT
for v in unset 1 0; do
  for wl in cfg5 cfg2; do
    if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
    MZ_BENCH_MIN_S=2 timeout -k 10 180 python bench.py --no-cpu-baseline --workload $wl --warmup 5 > gpurun_out/ka.json 2> gpurun_out/ka.err || { echo fail; tail -3 gpurun_out/ka.err; exit 1; }
    python -c "
import json;d=json.loads(open('gpurun_out/ka.json').read().strip().splitlines()[-1]);print('HIP_FORCE_DEV_KERNARG=$v', d['config']['name'], 'ms/step %.4f' % d['ms_per_step'])"
  done
done
