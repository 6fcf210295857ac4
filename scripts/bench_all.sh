#!/bin/bash
# Runs the benchmark workloads back to back on the GPU box; outputs under gpurun_out/.
set -e
tag=${1:-run}
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/${tag}_cfg2.json 2> gpurun_out/${tag}_cfg2.err
timeout -k 10 200 python bench.py --workload cfg4s --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_cfg4s.json 2> gpurun_out/${tag}_cfg4s.err
timeout -k 10 200 python bench.py --workload cfg3h --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_cfg3h.json 2> gpurun_out/${tag}_cfg3h.err
timeout -k 10 200 python bench.py --workload cfg5h --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_cfg5h.json 2> gpurun_out/${tag}_cfg5h.err
python - <<PY
import json
for w in ('cfg2','cfg4s','cfg3h','cfg5h'):
    j=json.load(open(f'gpurun_out/${tag}_{w}.json'))
    print(w, f"{j['value']:.3e} sims/s  {j['ms_per_step']:.3f} ms/step  nv/s={j['node_visits_per_sec']:.3e}  depth={j['mean_leaf_depth']:.2f}  roof={j['roofline']['frac']:.4f} kern_ms={j['roofline']['kernel_ms']:.3f}")
PY
