"""Summarises the shader-sequencer counter passes of scripts/profile_sq.sh (gpurun_out/<tag>_<workload>/) into
profiles/<round>_sq_counters.txt: mean per launch of the dominant kernel and the shares of wave cycles.
Usage: python scripts/sq_summary.py <tag> <round-name>"""
import csv, glob, sys
from collections import defaultdict

tag, rnd = sys.argv[1], sys.argv[2]
DOMINANT = {'cfg3': 'k_search_gcache', 'cfg2': 'k_search_reg', 'cfg4s': 'k_search_reg', 'cfg3h': 'k_search_hash'}
lines = ['rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS',
         '(scripts/profile_sq.sh; bench.py --steps 3 --warmup 1; mean per launch of the dominant kernel; SQ_WAVE_CYCLES counts units of 4 clocks)', '']
for w, kern in DOMINANT.items():
    files = glob.glob(f'gpurun_out/{tag}_{w}/**/*counter_collection.csv', recursive=True)
    if not files:
        continue
    acc, name = defaultdict(list), None
    for row in csv.DictReader(open(files[0])):
        if kern in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
            name = row['Kernel_Name'].split('(')[0]
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    lines.append(f'{w} {name}')
    for k in sorted(m):
        lines.append(f'    {k:<22} {m[k]:.4g}')
    wc = m['SQ_WAVE_CYCLES']
    lines.append(f"    share of wave cycles: parked on waitcnt/barrier {m['SQ_WAIT_ANY'] / wc:.2f}, issue stall "
                 f"{m['SQ_WAIT_INST_ANY'] / wc:.2f}, issuing {m['SQ_ACTIVE_INST_ANY'] / wc:.2f}; VALU instructions per wave cycle "
                 f"{m['SQ_INSTS_VALU'] / wc:.3f}")
    lines.append('')
open(f'profiles/{rnd}_sq_counters.txt', 'w').write('\n'.join(lines))
print('\n'.join(lines))
