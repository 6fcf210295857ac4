"""Turns the rocprofv3 outputs of scripts/profile_round.sh (under gpurun_out/) into the summaries kept
under profiles/: per-workload kernel statistics (csv copied as is) and HBM bytes per launch of the
dominant kernel.  Usage: python scripts/pmc_summary.py <tag> <round-name>"""
import csv, glob, json, os, shutil, sys
sys.path.insert(0, '.')
from muzero_amd.build import source_fingerprint

tag, rnd = sys.argv[1], sys.argv[2]
DOMINANT = {'cfg3': 'k_search_gcache', 'cfg2': 'k_search_reg', 'cfg4s': 'k_search_reg', 'cfg3h': 'k_search_hash',
            'cfg5': 'k_net_level_rows', 'cfg5d': 'k_net_level_rows'}
NOTE = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py (5 steps), mean over the '
        'kernel launches; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (counters are in KiB; gfx950 tallies 128-byte '
        'read requests at 64 bytes).  Calibrated on this access pattern with k_to_act of the same runs: 8192 root '
        'records of 128 B plus 128 KB of per-tree inputs = 1.15 MB read, FETCH_SIZE 573 KB; 360 KB written, '
        'WRITE_SIZE 352 KB')
PER_SEARCH = {'cfg5': 'k_stage_inputs', 'cfg5d': 'k_stage_inputs'}   # launched once per search
traffic, raw = {}, {}
st = glob.glob(f'gpurun_out/{tag}_stats_default/**/*kernel_stats.csv', recursive=True)
if st:
    shutil.copy(st[0], f'profiles/{rnd}_default_command_kernel_stats.csv')
    shutil.copy(f'gpurun_out/{tag}_stats_default.json', f'profiles/{rnd}_default_command_bench_under_rocprof.json')
for w, kern in DOMINANT.items():
    st = glob.glob(f'gpurun_out/{tag}_stats_{w}/**/*kernel_stats.csv', recursive=True)
    if st:
        shutil.copy(st[0], f'profiles/{rnd}_{w}_kernel_stats.csv')
    if os.path.exists(f'gpurun_out/{tag}_stats_{w}.json'):
        shutil.copy(f'gpurun_out/{tag}_stats_{w}.json', f'profiles/{rnd}_{w}_bench_under_rocprof.json')
    vals = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        files = glob.glob(f'gpurun_out/{tag}_pmc_{w}_{c}/**/*counter_collection.csv', recursive=True)
        if not files:
            continue
        xs, name, total, searches = [], None, 0.0, 0
        for row in csv.DictReader(open(files[0])):
            if row['Counter_Name'] != c:
                continue
            if 'mz::' in row['Kernel_Name']:
                total += float(row['Counter_Value'])
            if w in PER_SEARCH and PER_SEARCH[w] in row['Kernel_Name']:
                searches += 1
            if kern in row['Kernel_Name']:
                xs.append(float(row['Counter_Value']))
                name = row['Kernel_Name'].split('(')[0]
        if w in PER_SEARCH and searches:
            # a search of many launches: all of the engine's kernels, per search (bench.py prices a whole search)
            vals[c] = (total / searches, searches, f'all mz:: kernels of one search ({name} dominant)')
        elif xs:
            vals[c] = (sum(xs) / len(xs), len(xs), name)
    raw[w] = vals
    if len(vals) == 2:
        f, wr = vals['FETCH_SIZE'][0], vals['WRITE_SIZE'][0]
        traffic[w] = {'kernel': vals['FETCH_SIZE'][2], 'fetch_size_kb': f, 'write_size_kb': wr,
                      'traffic_bytes_per_launch': (2 * f + wr) * 1024, 'method': NOTE,
                      'kernel_sources': source_fingerprint()}
json.dump(raw, open(f'profiles/{rnd}_pmc_hbm_traffic_raw.json', 'w'), indent=1)
# workloads not profiled in this run keep their earlier entries (with the fingerprint of the sources they were measured on)
try:
    old = json.load(open('profiles/hbm_traffic.json'))
except OSError:
    old = {}
old.update(traffic)
traffic = old
json.dump(traffic, open('profiles/hbm_traffic.json', 'w'), indent=1)
print(json.dumps(traffic, indent=1))
