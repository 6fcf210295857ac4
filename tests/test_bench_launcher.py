"""CPU: `bench.py --gpus 2` started plainly (no torchrun) starts two ranks itself, they form a gloo process group,
exchange the per-move records, and rank 0 prints one line with n_gpus == 2.  The search step is stubbed
(MUZERO_AMD_BENCH_STUB=1), everything around it is the code the GPU run uses."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, drop=('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(MUZERO_AMD_BENCH_STUB='1', MUZERO_AMD_DIST_BACKEND='gloo')
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + args, env=env, capture_output=True,
                          text=True, timeout=240)


def test_plain_start_with_two_gpus_spawns_two_ranks():
    r = _run(['--gpus', '2', '--steps', '3', '--warmup', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                       # only rank 0 prints
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1
    assert out['config']['gathered_rows'] == 64 + 65   # both ranks' (uneven) shards reached rank 0
    assert out['scaling'] == 'weak' and out['data'].startswith('stub')


def test_world_size_that_disagrees_with_gpus_is_an_error():
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'], {'WORLD_SIZE': '1', 'RANK': '0'}, drop=())
    assert r.returncode != 0 and 'WORLD_SIZE' in r.stderr


def test_single_rank_stub_line():
    r = _run(['--steps', '2', '--warmup', '0'])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert out['n_gpus'] == 1
