"""Summarises the matrix-core counter pass of scripts/profile_mfma.sh (gpurun_out/<tag>_<workload>/) into
profiles/<round>_mfma_counters.txt: MFMA instructions, flops and busy cycles per launch of every kernel that issues
MFMA, and the utilisation they imply.  Usage: python scripts/mfma_summary.py <tag> <round-name>"""
import csv, glob, json, sys
from collections import defaultdict

tag, rnd = sys.argv[1], sys.argv[2]
PEAK_TF, SIMDS, FLOP_PER_MFMA, BUSY_PER_MFMA = 157.3, 1024, 2 * 16 * 16 * 4, 32
lines = ['rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES',
         '(scripts/profile_mfma.sh; bench.py --steps 3 --warmup 1; mean per launch; v_mfma_f32_16x16x4_f32 = 2048 flop, 32 busy cycles; '
         'durations from the kernel trace of the same run; peak 157.3 TFLOP/s float32 dense, 1024 SIMDs)', '']
for w in ('cfg3', 'cfg5'):
    cc = glob.glob(f'gpurun_out/{tag}_{w}/**/*counter_collection.csv', recursive=True)
    kt = glob.glob(f'gpurun_out/{tag}_{w}/**/*kernel_trace.csv', recursive=True)
    if not cc or not kt:
        continue
    dur = defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        dur[r['Kernel_Name'].split('(')[0]].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    try:
        bench = json.loads(open(f'gpurun_out/{tag}_{w}.json').read().strip().split('\n')[-1])
        lines.append(f"{w}: {bench['config']['workload']}")
    except Exception:
        lines.append(w)
    for k, v in acc.items():
        n = sum(v.get('SQ_INSTS_VALU_MFMA_F32', [0])) / max(1, len(v.get('SQ_INSTS_VALU_MFMA_F32', [0])))
        if n <= 0 or k not in dur or 'mz::' not in k:   # the engine's kernels only (torch's root inference also issues MFMA)
            continue
        t = sum(dur[k]) / len(dur[k]) * 1e-9
        busy = sum(v['SQ_VALU_MFMA_BUSY_CYCLES']) / len(v['SQ_VALU_MFMA_BUSY_CYCLES'])
        tf = n * FLOP_PER_MFMA / t / 1e12
        lines.append(f'    {k}: {len(dur[k])} launches, {t * 1e6:.1f} us (under the counter pass), {n:.4g} MFMA instructions = '
                     f'{n * FLOP_PER_MFMA / 1e9:.3g} GFLOP executed -> {tf:.1f} TFLOP/s = {tf / PEAK_TF:.3f} of peak; '
                     f'SQ_VALU_MFMA_BUSY_CYCLES {busy:.4g} = {busy / SIMDS:.4g} per SIMD')
    lines.append('')
open(f'profiles/{rnd}_mfma_counters.txt', 'w').write('\n'.join(lines))
print('\n'.join(lines))
