"""The measurement table of DESIGN.md, generated from a bench-lines file (one bench.py JSON line per workload) so that
the document and the recorded lines cannot disagree.

    python scripts/design_table.py profiles/round3_bench_lines.jsonl            # prints the table
    python scripts/design_table.py profiles/round3_bench_lines.jsonl --write    # rewrites it inside DESIGN.md

DESIGN.md holds the table between the markers <!-- bench-table:begin FILE --> and <!-- bench-table:end -->."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN = re.compile(r'<!-- bench-table:begin (\S+) -->')
END = '<!-- bench-table:end -->'


def si(x):
    for div, suffix in ((1e9, ' G'), (1e6, ' M'), (1e3, ' k')):
        if abs(x) >= div:
            return f'{x / div:.3g}{suffix}'
    return f'{x:.3g}'


def table(path):
    rows = ['| `--workload` | what one step is | ms/step | simulations/s | node-visits/s | dominant kernel(s) | HBM: algorithmic GB/s (frac of 8 TB/s), measured bytes / algorithmic | MFMA frac of 157.3 TF: executed flops (dense-equivalent flops) | CPU checker on the same workload |',
            '|---|---|---|---|---|---|---|---|---|']
    for line in open(path):
        line = line.strip()
        if not line:
            continue
        d = json.loads(line)
        r = d['roofline']
        alg = r.get('algorithmic_bytes_per_launch') or 0
        hbm = f"{r['achieved']:.0f} ({r['frac']:.3f})"
        hbm += f", {r['traffic'] / alg:.2f}x" if r.get('traffic') and alg else ', -'
        mf = d.get('roofline_mfma')
        cpu = d.get('cpu_baseline')
        extra = f"; {si(d['env_steps_per_sec'])} env-steps/s" if d.get('env_steps_per_sec') else ''
        rows.append('| `%s` | %s | %.3f | %s | %s%s | %s (%.3f ms) | %s | %s | %s |' % (
            d['config']['name'], d['config']['workload'], d['ms_per_step'], si(d['value']), si(d['node_visits_per_sec']), extra,
            r['kernel'], r['kernel_ms'], hbm,
            (f"{mf['frac']:.3f}" + (f" ({mf['dense_equivalent']['frac']:.3f})" if 'dense_equivalent' in mf else '')) if mf else '-',
            f"{si(cpu['value'])} sims/s on {cpu['cores']} threads ({cpu['kind']})" if cpu else '-'))
    return '\n'.join(rows)


def rewrite(text):
    """DESIGN.md with every marked table regenerated from the file its marker names."""
    out, pos = [], 0
    for m in BEGIN.finditer(text):
        end = text.index(END, m.end())
        out.append(text[pos:m.end()] + '\n' + table(os.path.join(ROOT, m.group(1))) + '\n')
        pos = end
    return ''.join(out) + text[pos:]


if __name__ == '__main__':
    if '--write' in sys.argv:
        p = os.path.join(ROOT, 'DESIGN.md')
        new = rewrite(open(p).read())
        open(p, 'w').write(new)
    else:
        print(table(sys.argv[1]))
