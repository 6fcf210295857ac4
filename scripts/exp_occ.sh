export MZ_BENCH_MIN_S=1 MZ_BENCH_SIMS=16
run() { name=$1; shift; env "$@" timeout -k 10 120 python bench.py --workload cfg3 --steps 50 --warmup 3 --no-cpu-baseline > gpurun_out/occ_$name.json 2>gpurun_out/occ_$name.err || return 1; }
run tm32 MZ_X=1 && run tm16w2 MZ_SEARCH_TM=16 && run tm16w3 MZ_SEARCH_TM=16 MZ_SEARCH_WPE=3
python - <<PY
import json
for n in ("tm32","tm16w2","tm16w3"):
    try:
        l=[x for x in open(f"gpurun_out/occ_{n}.json") if x.startswith("{")][-1]; d=json.loads(l); print(n, round(d["ms_per_step"],4), round(d["roofline"]["kernel_ms"],4))
    except Exception as e: print(n, "ERR", e)
PY
