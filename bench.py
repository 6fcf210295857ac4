"""Benchmark of the batched MCTS search (BASELINE.json metric: MCTS simulations/sec).

    python bench.py --gpus N --steps K --warmup W [--workload cfg1|cfg2|cfg3|cfg3b|cfg3h|cfg3s|cfg3t|cfg4|cfg4s|cfg5|cfg5d|cfg5h|cfg5s|cfg5t]

One process per GPU.  Under a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE (torchrun) the process is
one rank; started plainly with --gpus N > 1 it starts the N ranks itself, as fresh child processes, before
anything in it has touched a GPU, and exits with their status.  A step is one full batched search over the
rank's trees with inputs resident in HBM: root inference + root preparation, all simulations with in-kernel
leaf evaluation (one launch), final action rule.  Weak scaling: every rank searches its own B trees; there is
no data-path collective inside a search (self-play gathers each move's records on rank 0).

K steps are timed as one block bracketed by barrier + synchronize; the block is repeated (same K, same bracketing)
until MIN_TIMED_S seconds of timed work have accumulated, and the figures are taken over all blocks (`timed_blocks`,
`timed_region_s` in the output line).  MZ_BENCH_MIN_S overrides the duration (profiling passes use short runs).

Default workload = BASELINE.json configs[2], the largest single-GPU configuration: TicTacToe, 16384 trees x
100 simulations, Gumbel sequential halving, random-init MuZeroResidualNetwork (1 block, 16 channels) evaluated
on the matrix cores inside the search kernel.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from muzero_amd import distributed as mzdist  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3  # dense float32 matrix peak of MI355X (no xf32 on gfx950)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def cartpole_net_config():
    return SimpleNamespace(network='mlp', observation_dim=[1, 1, 4], n_actions=2, n_stacked_observations=1,
                           encoding_size=8, fc_reward_layers=[16], fc_policy_layers=[16], fc_value_layers=[16],
                           fc_representation_layers=[], fc_hidden_state_layers=[16], support_limit=10)


def search_config(gumbel, A, S, players, gamma):
    return SimpleNamespace(simulations=S, n_actions=A, gamma=gamma, players=players, gumbel=gumbel,
                           max_considered_actions=16, gumbel_scale=1.0, c_visit=50, c_scale=1.0,
                           c_base=19625, c_init=1.25, dirichlet_alpha=0.25, exploration_frac=0.25)


def atari_like_net_config():
    return SimpleNamespace(network='mlp', observation_dim=[1, 1, 128], n_actions=18, n_stacked_observations=1,
                           encoding_size=256, fc_reward_layers=[256], fc_policy_layers=[256], fc_value_layers=[256],
                           fc_representation_layers=[256], fc_hidden_state_layers=[256], support_limit=300)


def tictactoe_net_config():
    return SimpleNamespace(network='resnet', observation_dim=[3, 3, 3], n_actions=9, n_stacked_observations=1,
                           blocks=1, channels=16, reduced_channels_reward=8, reduced_channels_policy=8,
                           reduced_channels_value=8, resnet_fc_reward_layers=[16], resnet_fc_policy_layers=[16],
                           resnet_fc_value_layers=[16], downsample=False, support_limit=1)


WORKLOADS = {
    # name: (description, gumbel, A, S, players, gamma, B per GPU, leaf evaluator)
    'cfg1': ('TicTacToe 1 tree x 25 sims PUCT, random-init ResNet, through the drop-in MCTS.search call with host inputs and outputs (BASELINE configs[0])', False, 9, 25, 2, 1, 1, 'dropin'),
    'cfg2': ('CartPole 256 envs x 50 sims PUCT, random-init MLP (BASELINE configs[1])', False, 2, 50, 1, 0.997, 256, 'mlp'),
    'cfg4s': ('CartPole 8192 envs x 50 sims PUCT search only, random-init MLP (configs[3] per-GPU shard)', False, 2, 50, 1, 0.997, 8192, 'mlp'),
    'cfg4': ('CartPole self-play 8192 envs/GPU x 50 sims PUCT, fused env step recording every move in a device move log, per-step gather of the records to rank 0 (BASELINE configs[3] per-GPU shard)', False, 2, 50, 1, 0.997, 8192, 'selfplay'),
    'cfg3': ('TicTacToe 16384 trees x 100 sims Gumbel, random-init ResNet (1 block, 16 ch), whole search in one launch, leaves evaluated on the matrix cores (BASELINE configs[2])', True, 9, 100, 2, 1, 16384, 'mfma'),
    'cfg3b': ('TicTacToe 16384 trees x 100 sims Gumbel, random-init ResNet, through the drop-in MCTS.search_batch call: root noise drawn from np.random per call, roots evaluated and searched on the device, actions / values / policies copied to the host', True, 9, 100, 2, 1, 16384, 'dropin_batch'),
    'cfg3t': ('TicTacToe 16384 trees x 100 sims Gumbel, random-init ResNet, leaves evaluated by torch ops (comparison)', True, 9, 100, 2, 1, 16384, 'step'),
    'cfg3s': ('TicTacToe 16384 trees x 100 sims Gumbel, random-init ResNet, one leaf-evaluation launch + one backup/selection launch per simulation (comparison)', True, 9, 100, 2, 1, 16384, 'mfma_step'),
    'cfg3h': ('TicTacToe-shaped 16384 trees x 100 sims Gumbel, synthetic hash network (tree kernels only)', True, 9, 100, 2, 1, 16384, 'hash'),
    'cfg5': ('Atari-shaped 1024 trees x 200 sims PUCT, random-init MLP (hidden 256, 18 actions, 601-bin supports), whole search in one call: per simulation one matrix-core launch per network level and one tree launch (BASELINE configs[4] per-GPU shard)', False, 18, 200, 1, 0.997, 1024, 'mfma'),
    'cfg5d': ('Atari-shaped 1024 trees x 200 sims PUCT, hidden-256 MLP with a peaked policy head (deep trees), whole search in one call', False, 18, 200, 1, 0.997, 1024, 'mfma'),
    'cfg5s': ('Atari-shaped 1024 trees x 200 sims PUCT, random-init MLP (hidden 256), one call per simulation: a launch per network level, epilogue, backup + next selection (comparison)', False, 18, 200, 1, 0.997, 1024, 'mfma_step'),
    'cfg5t': ('Atari-shaped 1024 trees x 200 sims PUCT, random-init MLP (hidden 256), leaves evaluated by torch ops (comparison)', False, 18, 200, 1, 0.997, 1024, 'step5'),
    'cfg5h': ('Atari-shaped 1024 trees x 200 sims PUCT, hidden 256, 18 actions, synthetic hash network', False, 18, 200, 1, 0.997, 1024, 'hash'),
}


def algorithmic_bytes(node_visits_total, B, S, A, H, gumbel):
    """HBM bytes the tree work of one search needs, from the per-unit figures of the design:
    per selection level 36 + 6w + A(4 + p + 2w), per expansion 8H + pA + 20 + 3w, with the stored
    widths w = 8 (values/rewards) and p = 8 (PUCT priors) / 4 (Gumbel priors).  The number of
    selection levels is sum over simulations of the leaf depth = node_visits - B*S."""
    w, p = 8, (4 if gumbel else 8)
    levels = node_visits_total - B * S
    return levels * (36 + 6 * w + A * (4 + p + 2 * w)) + B * S * (8 * H + p * A + 20 + 3 * w)


DEFAULT_STEPS = {'cfg1': 200, 'cfg2': 2000, 'cfg3': 100, 'cfg3b': 50, 'cfg3h': 200, 'cfg3t': 3, 'cfg3s': 20, 'cfg4': 500, 'cfg4s': 500,
                 'cfg5': 50, 'cfg5d': 50, 'cfg5s': 20, 'cfg5h': 200, 'cfg5t': 5}
MIN_TIMED_S = 8.0     # repeat the block of K steps until this much timed work has accumulated (the GPU leg of the
                      # default command is then as long as its CPU-baseline leg)
MAX_BLOCKS = 2000


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='cfg3', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--lanes', type=int, default=0)
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = DEFAULT_STEPS[args.workload]
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        ap.error('--gpus and --steps must be positive, --warmup non-negative')
    return args


def launch_ranks(n: int) -> int:
    """Starts n ranks of this script as child processes (one per GPU) and waits for them.  Called before this
    process has made any GPU call: the children are fresh processes, nothing is re-executed in place.
    Rank 0's standard output is this process's; the exit status is the first non-zero child status."""
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    status = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0 and status == 0:
                status = rc
                for q in pending:           # a rank failed: the others would wait in a collective for ever
                    procs[q].terminate()
        time.sleep(0.05)
    return status


def main():
    args = parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    rank, world = mzdist.init()
    if world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}', file=sys.stderr)
        sys.exit(2)
    if os.environ.get('MUZERO_AMD_BENCH_STUB') == '1':
        return run_stub(args, rank, world)
    _, local_rank, _ = mzdist.env_rank_world()
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dev = torch.device('cuda', torch.cuda.current_device())

    from muzero_amd.engine import SearchEngine
    from muzero_amd.network import MuZeroNetwork
    from muzero_amd.packing import pack_network
    from muzero_amd import hashnet

    desc, gumbel, A, S, players, gamma, B, leaf = WORKLOADS[args.workload]
    if os.environ.get('MZ_BENCH_TREES'):   # experiments only: another batch size (the line's config says so)
        B = int(os.environ['MZ_BENCH_TREES'])
        desc += f' [experiment: {B} trees]'
    if os.environ.get('MZ_BENCH_SIMS'):    # experiments only: another simulation count
        S = int(os.environ['MZ_BENCH_SIMS'])
        desc += f' [experiment: {S} simulations]'
    for switch in ('MZ_PUCT_CACHE', 'MZ_SEARCH_CHAIN', 'MZ_SEARCH_CACHED', 'MZ_SEARCH_TM', 'MZ_IDX16'):
        if os.environ.get(switch) is not None:   # kernel-selection switches of the library: the line names them
            desc += f' [experiment: {switch}={os.environ[switch]}]'
    leaf_macs = None
    root_on_device = False
    cfg = search_config(gumbel, A, S, players, gamma)
    rng = np.random.default_rng(1 + rank)
    tree_id0 = rank * B
    packed = None
    sp = None
    gatherer = None
    net = None
    if leaf == 'dropin':
        sys.path.insert(0, os.path.join(REPO, 'muzero_amd'))
        from mcts.mcts import MCTS  # the drop-in package, imported under the reference's name
        cfg.dirichlet_alpha = 0.2
        net = MuZeroNetwork(tictactoe_net_config(), seed=0)
        H = net.H
        searcher = MCTS(cfg)
        obs0 = np.stack([np.zeros((3, 3)), np.zeros((3, 3)), np.full((3, 3), -1.0)])
        enc = lambda a: np.eye(9)[a].reshape(3, 3)  # noqa: E731
        np.random.seed(0)
        searcher.search(net, obs0, np.ones(9), enc, 0, 1.0)
        eng = searcher.last_engine
    elif leaf == 'dropin_batch':
        sys.path.insert(0, os.path.join(REPO, 'muzero_amd'))
        from mcts.mcts import MCTS  # the drop-in package, imported under the reference's name
        net = MuZeroNetwork(tictactoe_net_config(), seed=0)
        H = net.H
        searcher = MCTS(cfg)
        obs = torch.as_tensor(tictactoe_roots(rng, B).reshape(B, 27), device=dev)
        legal_h, to_play_h = np.ones((B, A)), np.zeros(B, np.int32)
        np.random.seed(0)
        searcher.search_batch(net, obs, legal_h, to_play_h, 1.0, tree_id0=tree_id0)
        eng = searcher.last_engine
    elif leaf == 'selfplay':
        from muzero_amd.self_play import BatchedSelfPlay
        cfg.max_moves = 500
        net = MuZeroNetwork(cartpole_net_config(), seed=0)
        packed = pack_network(net)
        H = packed.H
        sp = BatchedSelfPlay(cfg, 'cartpole', net, B, seed=rank, tree_id0=tree_id0, device=dev)
        eng = sp.engine
        # every move's records (action, root value, policy, observation, reward, done, to_play) to rank 0
        gatherer = mzdist.StepGather(B, {'probs': A, 'obs_before': sp.envs.obs_size}, device=dev, dst=0)
    elif leaf == 'mlp':
        torch.manual_seed(0)
        net = MuZeroNetwork(cartpole_net_config(), seed=0)
        packed = pack_network(net)
        H = packed.H
        eng = SearchEngine(cfg, H, B, seed=0, lanes_per_tree=args.lanes)
        eng.set_network(packed)
        obs = torch.as_tensor(rng.uniform(-0.05, 0.05, size=(B, 4)).astype(np.float32), device=dev)
    elif leaf in ('mfma', 'mfma_step', 'step5') and A == 18:
        from muzero_amd.packing import pack_network_mfma
        net = MuZeroNetwork(atari_like_net_config(), seed=0)
        if args.workload == 'cfg5d':   # one action preferred by a wide margin: searches run down long chains
            sd = {k: v.clone() for k, v in net.state_dict().items()}
            sd[sorted(k for k in sd if 'policy_network' in k and k.endswith('bias'))[-1]][7] += 6.0
            net = MuZeroNetwork(atari_like_net_config(), sd)
        net = net.to(dev)
        H = net.H
        eng = SearchEngine(cfg, H, B, seed=0, lanes_per_tree=args.lanes)
        pm_io = pack_network_mfma(net)
        leaf_macs = (pm_io[0].macs, pm_io[0].padded_macs)
        eng.set_network_mfma(*pm_io)
        if leaf in ('mfma', 'mfma_step'):
            from muzero_amd.packing import pack_root_network_mfma
            eng.set_root_network_mfma(*pack_root_network_mfma(net))
            root_on_device = True
        obs = torch.as_tensor(rng.standard_normal((B, 128)).astype(np.float32), device=dev)
    elif leaf in ('step', 'mfma', 'mfma_step'):
        from muzero_amd.packing import pack_network_mfma
        net = MuZeroNetwork(tictactoe_net_config(), seed=0).to(dev)
        H = net.H
        eng = SearchEngine(cfg, H, B, seed=0, lanes_per_tree=args.lanes)
        pm_io = pack_network_mfma(net)
        leaf_macs = (pm_io[0].macs, pm_io[0].padded_macs)
        eng.set_network_mfma(*pm_io)
        if leaf in ('mfma', 'mfma_step'):   # the roots, too, are evaluated by the engine (mz_initial_inference_mfma)
            from muzero_amd.packing import pack_root_network_mfma
            eng.set_root_network_mfma(*pack_root_network_mfma(net))
            root_on_device = True
        obs = torch.as_tensor(tictactoe_roots(rng, B).reshape(B, 27), device=dev)
    else:
        H = 144 if A == 9 else 256
        eng = SearchEngine(cfg, H, B, seed=0, lanes_per_tree=args.lanes)
        hp = hashnet.HashNetParams(A, H, 4, 1, 1)
        codes = rng.integers(0, 2**32, size=B, dtype=np.uint64).astype(np.uint32)
        hid = np.zeros((B, H), np.float32)
        hid[:, 0], hid[:, 1] = codes & 0xFFFF, codes >> 16
        root_hidden = torch.as_tensor(hid, device=dev)
        root_logits = torch.as_tensor(rng.standard_normal((B, A)).astype(np.float32), device=dev)
        root_value = torch.zeros(B, dtype=torch.float64, device=dev)
    legal = torch.ones((B, A), dtype=torch.uint8, device=dev)
    to_play = torch.zeros(B, dtype=torch.int32, device=dev)
    dirichlet = None if gumbel else torch.as_tensor(rng.dirichlet([0.25] * A, size=B), device=dev)
    gnoise = torch.as_tensor(rng.gumbel(size=(B, A)), device=dev) if gumbel else None
    u = torch.as_tensor(rng.random(B), device=dev)
    temperature = 1.0

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    # 2-action PUCT trees (<= 64 nodes) run with the tree in registers (csrc/search_reg.h);
    # that kernel also runs the final action rule, so search + rule are one launch
    reg_tree = (leaf == 'mlp' and not gumbel and A == 2 and S + 1 <= 64 and
                os.environ.get('MZ_TREE_REG', '1') != '0')
    one_call = os.environ.get('MZ_BENCH_CALLS', 'act')   # 'move' | 'act' | 'separate': how many launches make a step

    gathered = [0]   # rows of the last per-move gather on this rank (all ranks' environments on the replay rank)

    def step(i=None):
        if leaf == 'dropin':
            if i is not None:
                ev[i][0].record()
            out = searcher.search(net, obs0, np.ones(9), enc, 0, 1.0)
            if i is not None:
                ev[i][1].record()
            return out
        if leaf == 'dropin_batch':
            if i is not None:
                ev[i][0].record()
            out = searcher.search_batch(net, obs, legal_h, to_play_h, 1.0, tree_id0=tree_id0)
            if i is not None:
                ev[i][1].record()
            return out
        if leaf == 'selfplay':
            if i is not None:
                ev[i][0].record()
            out = sp.step(temperature)
            if i is not None:
                ev[i][1].record()
            full = gatherer.gather(out)  # this move's records to the replay rank (a no-op on one rank)
            if full is not None:
                gathered[0] = int(full['action'].shape[0])
            return out
        if leaf == 'mlp' and reg_tree and one_call == 'move':   # roots, simulations, final action rule: one call
            if i is not None:
                ev[i][0].record()
            out = eng.search_move(obs, legal, to_play, dirichlet, 0.25, gnoise, tree_id0, temperature, u)
            if i is not None:
                ev[i][1].record()
            return out
        if leaf == 'mlp' and reg_tree and one_call == 'act':    # roots; simulations + final action rule
            eng.begin_net(obs, legal, to_play, dirichlet, 0.25, gnoise, tree_id0=tree_id0)
            if i is not None:
                ev[i][0].record()
            out = eng.search_net_act(temperature, u)
            if i is not None:
                ev[i][1].record()
            return out
        if leaf == 'mlp':
            eng.begin_net(obs, legal, to_play, dirichlet, 0.25, gnoise, tree_id0=tree_id0)
        elif leaf in ('step', 'mfma', 'mfma_step', 'step5'):
            lg0, hid0, val0 = eng.initial_inference_mfma(obs) if root_on_device else net.initial_flat(obs)
            prior = eng.root_prepare(lg0, legal, dirichlet, 0.25)
            eng.reset(prior, val0, hid0, to_play, legal, gnoise, tree_id0=tree_id0)
        else:
            prior = eng.root_prepare(root_logits, legal, dirichlet, 0.25)
            eng.reset(prior, root_value, root_hidden, to_play, legal, gnoise, tree_id0=tree_id0)
        if i is not None:
            ev[i][0].record()
        if leaf == 'mlp':
            eng.search_net()
        elif leaf == 'mfma':
            eng.search_mfma()
        elif leaf == 'mfma_step':
            eng.select(want_hidden=False)
            for _ in range(S):
                eng.expand_backup_net(select_next=True)
        elif leaf in ('step', 'step5'):
            for _ in range(S):
                parent, action, phid = eng.select()
                lg, nh, val, rew = net.recurrent_flat(phid, net.encode_actions(action))
                eng.expand_backup(lg, val, rew, nh)
        else:
            eng.search_hash(4, 1, 1)
        if i is not None:
            ev[i][1].record()
        return eng.to_act(temperature, u)

    games_recorded = 0
    if sp is not None and args.warmup:
        sp.record(args.warmup)
    for _ in range(args.warmup):
        step()
    rdev = mzdist.reduce_device(dev)
    elapsed, kern_ms_all, blocks = 0.0, [], 0
    while True:
        torch.cuda.synchronize()
        mzdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if sp is not None:
            sp.record(args.steps)      # every move of the block goes to the device move log
        for i in range(args.steps):
            step(i)
        if sp is not None:             # the block's finished games, listed on the device (part of the timed region)
            try:
                games_recorded += len(sp.harvest())
            except ValueError:
                pass
        torch.cuda.synchronize()
        mzdist.barrier()
        block = mzdist.max_over_ranks(time.perf_counter() - t0, rdev)   # the same number on every rank
        elapsed += block
        blocks += 1
        kern_ms_all += [a.elapsed_time(b) for a, b in ev]
        if elapsed >= float(os.environ.get('MZ_BENCH_MIN_S', MIN_TIMED_S)) or blocks >= MAX_BLOCKS:
            break
    n_steps = blocks * args.steps

    nv, _ = eng.stats()
    node_visits = int(nv.sum().item())
    kern_ms = float(np.mean(kern_ms_all))
    alg = algorithmic_bytes(node_visits, B, S, A, H, gumbel)
    achieved = alg / (kern_ms * 1e-3) / 1e9
    sims = world * B * S * n_steps
    nv_all = mzdist.sum_over_ranks(node_visits, rdev)

    # HBM bytes per launch from the PMC counters, measured offline with rocprofv3 (profiles/hbm_traffic.json).  An entry
    # belongs to the kernel sources it was measured on: with other sources the line carries no traffic figure.
    traffic, traffic_note = None, None
    try:
        from muzero_amd.build import source_fingerprint
        with open(os.path.join(REPO, 'profiles', 'hbm_traffic.json')) as f:
            entry = json.load(f).get(args.workload)
        if entry is not None and '[experiment' not in desc:   # an experiment runs other sizes or other kernels than the entry's
            if entry.get('kernel_sources') == source_fingerprint():
                traffic = entry.get('traffic_bytes_per_launch')
            else:
                traffic_note = ('profiles/hbm_traffic.json was measured on other kernel sources (%s, now %s): run '
                                'scripts/profile_round.sh + scripts/pmc_summary.py again'
                                % (entry.get('kernel_sources'), source_fingerprint()))
    except OSError:
        pass
    out = {
        'metric': 'MCTS simulations/sec', 'value': sims / elapsed, 'unit': 'simulations/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / n_steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
        'data': 'synthetic', 'config': {'workload': desc, 'name': args.workload, 'trees_per_gpu': B, 'simulations': S,
                                        'n_actions': A, 'rule': 'gumbel' if gumbel else 'puct', 'leaf_eval': leaf,
                                        'lanes_per_tree': eng and int(args.lanes)},
        'timed_blocks': blocks, 'timed_region_s': elapsed,
        'node_visits_per_sec': nv_all * n_steps / elapsed,
        'mean_leaf_depth': node_visits / (B * S) - 1.0,
        'env_steps_per_sec': (world * B * n_steps / elapsed) if leaf == 'selfplay' else None,
        'recording': ({'where': 'device move log, written by k_act_env', 'games_recorded_rank0': games_recorded,
                       'moves_recorded_rank0': B * n_steps, 'gathered_rows_per_move_rank0': gathered[0]} if leaf == 'selfplay' else None),
        'roofline': {'bound': 'hbm', 'kernel': {'mlp': 'k_search_reg' if reg_tree else 'k_search_net', 'hash': 'k_search_hash', 'step': 'k_select+leaf evaluation+k_expand_backup',
                                'mfma': ('k_search_gcache' if gumbel and A == 9 else 'k_net_level_rows (one per network level) + k_leaf_backup_select, per simulation' if A > 16 else 'k_search_mfma'), 'mfma_step': ('k_stage_inputs + k_net_level_rows x levels + k_net_epilogue + k_backup_select, per simulation' if A > 16 else 'k_net_step+k_backup_select'), 'step5': 'k_select+leaf evaluation+k_expand_backup', 'dropin': 'MCTS.search (root inference on the level kernels, k_search_mfma, host copies)', 'dropin_batch': 'MCTS.search_batch (root noise from np.random, root inference on the level kernels, k_search_gcache, host copies)',
                                'selfplay': 'k_begin_net+k_search_reg+k_act_env'}[leaf],
                     'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                     'traffic': traffic, 'traffic_note': traffic_note, 'kernel_ms': kern_ms, 'algorithmic_bytes_per_launch': alg,
                     'limiter': LIMITER['mfma_levels'] if (leaf in ('mfma', 'mfma_step') and A > 16) else LIMITER.get(leaf, LIMITER['tree'])},
    }

    if leaf in ('mfma', 'mfma_step') and leaf_macs is not None:
        # leaves evaluated on the matrix cores: the lowered network's multiply-accumulates (unpadded) per leaf x leaves
        # per launch / the same launch time, against the dense float32 MFMA peak (v_mfma_f32_16x16x4_f32)
        # priced on the multiply-accumulates the matrix instructions execute (zero blocks of lowered convolutions are
        # skipped, the kept blocks are padded to 16 x 16); the dense-equivalent figure of the unlowered layers rides along
        tf = 2.0 * leaf_macs[1] * B * S / (kern_ms * 1e-3) / 1e12
        tf_dense = 2.0 * leaf_macs[0] * B * S / (kern_ms * 1e-3) / 1e12
        out['roofline_mfma'] = {'bound': 'mfma', 'achieved': tf, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                'frac': tf / MFMA_F32_PEAK_TFLOPS, 'executed_flop_per_leaf': 2 * leaf_macs[1],
                                'dense_equivalent': {'achieved': tf_dense, 'frac': tf_dense / MFMA_F32_PEAK_TFLOPS,
                                                     'flop_per_leaf': 2 * leaf_macs[0]},
                                'note': 'the launch also holds the tree operations of the same simulations'}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(cfg, net, leaf, A, S, H, B, gumbel)
    if rank == 0:
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        torch.distributed.destroy_process_group()


LIMITER = {
    'tree': 'not HBM: the shader-sequencer counters show the large-batch tree kernels at 70-85% of the VALU issue rate '
            '(float64 arithmetic in the reference\'s operation order); small batches are one wave per tree, bound by the '
            'latency of ~1000 dependent instructions per simulation',
    'mfma': 'per simulation a 32-tree block walks a cached decision table in LDS, runs the float32 matrix-core network of its '
            'tile (block-sparse weights streamed from L2), then the float64 epilogue, backup and re-decision of the path '
            'nodes (cheap scores with a certified argmax, the reference operations only without a certified lead); the '
            'shader-sequencer counters show waves parked on waits 47% and stalled on issue 25% of their cycles: neither '
            'HBM (measured traffic 2.0 TB/s) nor the matrix pipes (20% busy) saturate',
    'mfma_levels': 'a chain of three launches per simulation over only 1024 trees: every launch is bound by its own start-up, '
                   'memory round trips and drain (matrix instructions are 3.3 of a level launch\'s 13.7 us), and the chain '
                   'is serial by construction (the next selection needs this backup); the tree launch lasts as long as its '
                   'deepest tree (one wave per tree: epilogue, backup, re-decision of the path nodes by lane groups, a walk '
                   'down the cached PUCT decisions)',
}


def tictactoe_roots(rng, B):
    """Observations (planes: X stones, O stones, side to move) of B random TicTacToe positions."""
    boards = rng.integers(0, 3, size=(B, 9)) - 1  # random (not necessarily reachable) positions
    return np.stack([(boards == -1), (boards == 1), np.ones_like(boards) * -1], axis=1).astype(np.float32)


def run_stub(args, rank, world):
    """MUZERO_AMD_BENCH_STUB=1: everything around the search (rank start-up, process group, barriers, the
    max-over-ranks clock, the per-move gather, rank 0's output line) with an empty step and no GPU, so that the
    multi-rank plumbing can be tested on hosts without one.  The line says so in `data`."""
    B, A, OBS = 64 + rank, 2, 4   # uneven shards on purpose
    g = mzdist.StepGather(B, {'probs': A, 'obs_before': OBS}, device='cpu', dst=0)
    rec = dict(action=torch.zeros(B, dtype=torch.int32), value=torch.zeros(B, dtype=torch.float64),
               probs=torch.zeros((B, A), dtype=torch.float64), obs_before=torch.zeros((B, OBS)),
               reward=torch.zeros(B), done=torch.zeros(B, dtype=torch.uint8), to_play=torch.zeros(B, dtype=torch.int32))
    gathered = 0
    for _ in range(args.warmup):
        g.gather(rec)
    mzdist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = g.gather(rec)
        gathered = 0 if full is None else int(full['action'].shape[0])
    mzdist.barrier()
    elapsed = mzdist.max_over_ranks(time.perf_counter() - t0)
    total = int(mzdist.sum_over_ranks(B))
    if rank == 0:
        assert gathered == total, (gathered, total)
        print(json.dumps({'metric': 'MCTS simulations/sec', 'value': 0.0, 'unit': 'simulations/s', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
                          'data': 'stub (no search executed)', 'config': {'workload': 'stub', 'gathered_rows': gathered}}),
              flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def cpu_baseline(cfg, net, leaf, A, S, H, B, gumbel, budget_s=8.0):
    """The CPU checker (oracle/, a C port of the reference's search) timed on this box's host cores on a
    bounded sample of the SAME workload - same rule, simulations, action count and leaf evaluator (the lowered
    network executed as a dense program; the hash network for the hash workloads): batches of trees, one tree
    per OpenMP thread, repeated until the time budget is used."""
    from oracle import oracle as orc
    from muzero_amd import hashnet
    from muzero_amd.packing import pack_network
    threads = int(os.environ.get('OMP_NUM_THREADS', 0)) or (os.cpu_count() or 1)
    rng = np.random.default_rng(1)
    kind = 'hash network'
    if leaf in ('mlp', 'selfplay', 'mfma', 'step', 'mfma_step', 'step5', 'dropin', 'dropin_batch'):
        cpu_net = net.to('cpu') if hasattr(net, 'to') else net
        packed = pack_network(cpu_net)
        # a batch that takes on the order of a second: ~2 ns per multiply-accumulate and thread
        per_tree = S * max(packed.recurrent.macs, 1) * 2e-9
        n = int(min(B, max(threads, math.ceil(1.0 * threads / per_tree))))
        if leaf in ('mlp', 'selfplay'):
            obs = rng.uniform(-0.05, 0.05, size=(n, 4)).astype(np.float32)
        elif A == 9:
            obs = tictactoe_roots(rng, n).reshape(n, 27)
        else:
            obs = rng.standard_normal((n, 128)).astype(np.float32)
        with torch.no_grad():
            lg, hid, val = cpu_net.initial_flat(torch.as_tensor(obs))
        legal, tp = np.ones((n, A)), np.zeros(n, np.int32)
        di = None if gumbel else rng.dirichlet([0.25] * A, size=n)
        g = rng.gumbel(size=(n, A)) if gumbel else None
        run = lambda: orc.search_batch_net(cfg, packed, lg.numpy(), hid.numpy(), val.numpy(), legal, tp, dirichlet=di,  # noqa: E731
                                           frac=0.25, gumbel=g, seed=0, n_threads=threads)
        kind = 'the same lowered network as a dense float32 program'
    else:
        hp = hashnet.HashNetParams(A, H, 4, 1, 1)
        n = min(B, 2048)
        legal, tp = np.ones((n, A)), np.zeros(n, np.int32)
        di = None if gumbel else rng.dirichlet([0.25] * A, size=n)
        g = rng.gumbel(size=(n, A)) if gumbel else None
        codes = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
        run = lambda: orc.search_batch_hash(cfg, hp, H, codes, legal, tp, dirichlet=di, frac=0.25, gumbel=g, seed=0,  # noqa: E731
                                            n_threads=threads)
    run()
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        r = run()
        reps += 1
    dt = time.perf_counter() - t0
    return {'value': reps * n * S / dt, 'unit': 'simulations/s', 'cores': threads, 'kind': 'port',
            'sample': f'{reps} batches of {n} trees x {S} sims ({dt:.1f} s), leaves: {kind}; one tree per OpenMP thread',
            'node_visits_per_sec': reps * float(r.node_visits.sum()) / dt}


if __name__ == '__main__':
    main()
