"""Packs dense programs (muzero_amd/network.py) into the flat arrays the HIP kernels and the C
checker consume.

A program is scheduled into LEVELS: a step's level is one more than the deepest level among the
producers of its input and skip tensors, so all rows (output neurons) of a level can be computed
at once from the activation scratch.  Per level the weights are zero-padded to the level's widest
input (rounded up to 4) and stored as ``W4[k/4][row][4]``: lane `row` reads one 16-byte vector per
four inputs, consecutive lanes read consecutive vectors.  Tensor offsets in the scratch are multiples
of 4 floats so the inputs are read as 16-byte vectors too.

int32 meta:
  [0] n_levels   [1] scratch floats per tree   [2] total rows   [3] floats in data   [4] bias offset in data
  [5] index of the level table in meta          [6] index of the row table in meta    [7] reserved
  level table, 4 ints per level: n_rows, kmax, weight offset in data, first row
  row table, 4 ints per row:     input offset, output offset, activation (0 none, 1 relu, 2 elu), skip offset or -1
float32 data: per level W4[kmax/4][n_rows][4] (kmax a multiple of 4), then the biases of all rows.
Offsets of inputs/outputs are positions in a tree's activation scratch.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Sequence

import numpy as np

ACT = {'none': 0, 'relu': 1, 'elu': 2}


@dataclass
class PackedProgram:
    meta: np.ndarray          # int32
    data: np.ndarray          # float32
    slots: Dict[str, int]     # tensor name -> offset in the scratch
    sizes: Dict[str, int]     # tensor name -> floats
    macs: int                 # multiply-accumulates per evaluation (unpadded)

    @property
    def n_levels(self) -> int:
        return int(self.meta[0])

    @property
    def scratch(self) -> int:
        return int(self.meta[1])

    def weight_bytes(self) -> int:
        return int(self.data.nbytes + self.meta.nbytes)


def pack(steps: Sequence, inputs: Dict[str, int], aliases: Dict[str, tuple] = None) -> PackedProgram:
    """steps: DenseStep list; inputs: name -> size of every external input, laid out first in the
    scratch in the given order; aliases: name -> (base tensor, offset, size) views of inputs."""
    aliases = aliases or {}
    slots: Dict[str, int] = {}
    sizes: Dict[str, int] = {}
    level_of: Dict[str, int] = {}
    off = 0
    al = lambda x: (int(x) + 3) // 4 * 4  # noqa: E731  (16-byte alignment of every tensor)
    for name, size in inputs.items():
        slots[name], sizes[name], level_of[name] = off, int(size), 0
        off = al(off + int(size))
    for name, (base, o, size) in aliases.items():
        slots[name], sizes[name], level_of[name] = slots[base] + o, int(size), 0
    levels: List[list] = []
    for s in steps:
        N, K = s.W.shape
        if sizes[s.inp] != K:
            raise ValueError(f'step {s.out}: input {s.inp} has {sizes[s.inp]} floats, weight expects {K}')
        lv = 1 + max(level_of[s.inp], level_of[s.skip] if s.skip else 0)
        level_of[s.out] = lv
        slots[s.out], sizes[s.out] = off, int(N)
        off = al(off + int(N))
        while len(levels) < lv:
            levels.append([])
        levels[lv - 1].append(s)
    n_rows = sum(int(s.W.shape[0]) for s in steps)
    meta = [len(levels), off, n_rows, 0, 0, 8, 8 + 4 * len(levels), 0]
    level_tab, row_tab, blobs, biases = [], [], [], []
    w_off, row0, macs = 0, 0, 0
    for lv in levels:
        kmax = al(max(int(s.W.shape[1]) for s in lv))
        rows = sum(int(s.W.shape[0]) for s in lv)
        WT = np.zeros((kmax, rows), np.float32)
        r = 0
        for s in lv:
            N, K = s.W.shape
            WT[:K, r:r + N] = s.W.detach().cpu().numpy().T
            b = s.b.detach().cpu().numpy()
            for j in range(N):
                row_tab += [slots[s.inp], slots[s.out] + j, ACT[s.act], (slots[s.skip] + j) if s.skip else -1]
                biases.append(b[j])
            r += N
            macs += N * K
        # a row that reads fewer than kmax inputs multiplies the zero padding by whatever follows its
        # input in the scratch, so the scratch must extend kmax floats past every input offset
        level_tab += [rows, kmax, w_off, row0]
        blobs.append(WT.reshape(kmax // 4, 4, rows).transpose(0, 2, 1).reshape(-1))
        w_off += WT.size
        row0 += rows
    pad = max(level_tab[1::4], default=0)
    meta[1] = off + pad
    n_data = al(w_off + len(biases))  # padded so that what follows the data in LDS stays 16-byte aligned
    meta[3] = n_data
    meta[4] = w_off
    data = np.zeros(n_data, np.float32)
    if blobs:
        flat = np.concatenate(blobs + [np.asarray(biases, np.float32)])
        data[:flat.size] = flat
    return PackedProgram(np.asarray(meta + level_tab + row_tab, np.int32), data.astype(np.float32), slots, sizes, macs)


@dataclass
class PackedNetwork:
    """Everything the engine / checker need to evaluate a lowered MuZero network."""
    recurrent: PackedProgram   # inputs: hidden[H] + action[action_size]; dynamics and prediction merged
    represent: PackedProgram   # input: obs
    predict: PackedProgram     # input: hidden
    H: int
    A: int
    action_size: int
    obs_size: int
    support_limit: int
    norm_groups: int

    def io(self) -> np.ndarray:
        """int32[12]: scratch offsets of the recurrent program's inputs/outputs and the shape facts."""
        r = self.recurrent
        return np.asarray([r.slots['xa'], r.slots['xa'] + self.H, r.slots['next_raw'], r.slots['reward_logits'],
                           r.slots['policy_logits'], r.slots['value_logits'], self.H, self.A, self.action_size,
                           2 * self.support_limit + 1, self.norm_groups, self.obs_size], np.int32)


def pack_network(net) -> PackedNetwork:
    """net: muzero_amd.network.MuZeroDenseNetwork."""
    H, asz = net.H, net.action_size
    rec = pack(list(net.dyn.steps) + list(net.pred.steps), {'xa': H + asz}, {'hidden': ('xa', 0, H)})
    rep = pack(net.rep.steps, {'obs': net.obs_size})
    pred = pack(net.pred.steps, {'hidden': H})
    return PackedNetwork(rec, rep, pred, H, net.A, asz, net.obs_size, net.support_limit, net.norm_groups)


# ---------------------------------------------------------------------- MFMA layout
# For programs too large for LDS-resident weights (board ResNets, wide MLPs) a tile of trees keeps
# its ACTIVATIONS in LDS and streams the weights, pre-swizzled into v_mfma_f32_16x16x4_f32 operand
# order, from L2.  A step (one dense layer) is cut into 16-column tiles `ct` and 16-input blocks `k16`;
# only blocks that hold a non-zero weight are stored and multiplied (BLOCK-SPARSE): the dense matrix of
# a 3x3 convolution on a small board is 40 % zero blocks once its tensors are laid out position-major
# ([cell][channel], see `plane_major` below), because a cell only sees its neighbours.  Per (step, ct) job:
#   ctinfo[job] = (weight offset in floats, first entry in kidx, number of blocks)
#   kidx[...]   = the k16 indices of the job's blocks, ascending
#   Wfrag[block][lane 0..63][4]   with   col = ct*16 + (lane & 15),   k = k16*16 + (lane >> 4)*4 + j
# so one 16-byte load per lane feeds four MFMA k-steps (a wave reads 1 KiB contiguously) and the
# matching activations of a lane are four consecutive floats (one 16-byte LDS read).
# int32 step table, 12 ints per step:
#   level, N, K16 (= ceil(K/16)), in_off, out_off, skip_off (-1), act, w_off (floats, first job), b_off (floats),
#   n_col_tiles, first job in ctinfo, 0
# Work of a tile-resident block is scheduled on the host: within a level the (step, ct) jobs are dealt to the block's
# waves longest-first onto the least loaded wave, and a job of a 32-tree tile may be split into its two 16-row halves
# when that evens the waves out.  `sched` holds two schedules (8 waves = 32-tree tiles, 4 waves = 16-tree tiles):
#   sched[0:4] = offsets inside sched of wstart8, wdesc8, wstart4, wdesc4
#   wstartW[wave * (n_levels + 1) + l] .. [.. + l + 1] = the wave's jobs of level l (numbers of descriptors in wdescW);
#   a wave's jobs are contiguous over the levels (at most 64 per wave: a wave keeps its descriptors in registers)
#   wdescW[DESC_INTS * j ..] = one job, everything its wave needs without further table reads:
#     0 byte offset of its first weight block   1 number of blocks   2 mode (0: both 16-row halves, 1: rows 0..15,
#     2: rows 16..31)   3 input offset   4 first output column (floats in the activation row)   5 first residual
#     column or -1   6 activation   7 first bias entry   8 valid columns (1..16)   9 first entry in kidx
#     10..12 the first 12 block indices, one byte each
DESC_INTS = 13


@dataclass
class PackedMfma:
    steps: np.ndarray        # int32 [n_steps * 12]
    n_levels: int
    sched: np.ndarray        # int32, the two per-wave schedules
    wfrag: np.ndarray        # float32, the stored blocks
    bias: np.ndarray         # float32 (each step padded to 16 columns)
    stride: int              # activation floats per tree (4 x an odd number: conflict-free 16-byte operand reads)
    slots: Dict[str, int]
    macs: int                # unpadded multiply-accumulates per tree
    padded_macs: int         # what the MFMA tiles actually compute per tree (stored blocks x 256)
    ctinfo: np.ndarray = None   # int32 [n_jobs * 3]
    kidx: np.ndarray = None     # int32 [n_blocks]
    plane_major: int = 0        # 1: the hidden state inside the activation row is [cell][channel] (channels = the
                                # normalisation groups), not the [channel][cell] order of the stored hidden states
    dense_macs: int = 0         # what a dense evaluation of the same tiles would compute


def _assign_slots(steps, inputs, aliases, keep, reuse=True):
    """Offsets (in floats, multiples of 16) of every tensor in a tree's activation row, reusing the
    space of tensors that are dead: a tensor lives from the level that writes it (inputs: level 0) to
    the last level that reads it (`keep`: to the end).  Everything read or written at a level is
    disjoint, since the waves of a block work on different jobs of that level at the same time.
    `reuse` False: every tensor keeps a place of its own (the single-launch search that chains the levels
    across workgroups has no launch boundary between the last reader of a place and its next writer)."""
    al16 = lambda x: (int(x) + 15) // 16 * 16  # noqa: E731
    base_of = {name: b for name, (b, _, _) in aliases.items()}
    root = lambda n: base_of.get(n, n)  # noqa: E731
    sizes = {n: int(sz) for n, sz in inputs.items()}
    level_of = {n: 0 for n in inputs}
    last_use = {n: 0 for n in inputs}
    for s in steps:
        lv = 1 + max(level_of[root(s.inp)], level_of[root(s.skip)] if s.skip else 0)
        level_of[s.out], sizes[s.out] = lv, int(s.W.shape[0])
        last_use[s.out] = lv
        for r in (s.inp, s.skip):
            if r:
                last_use[root(r)] = max(last_use[root(r)], lv)
    for n in (list(last_use) if not reuse else keep):
        last_use[root(n)] = 1 << 30
    slots, live, end = {}, [], 0   # live: (offset, units of 16 floats, name)

    def place(name):
        nonlocal end
        need = al16(sizes[name])
        taken = sorted((o, o + n) for o, n, _ in live)
        pos = 0
        for lo, hi in taken:
            if lo - pos >= need:
                break
            pos = max(pos, hi)
        slots[name] = pos
        live.append((pos, need, name))
        end = max(end, pos + need)

    for n in inputs:
        place(n)
    for lv in range(1, max(level_of.values(), default=0) + 1):
        live[:] = [x for x in live if last_use[x[2]] >= lv]
        for s in steps:
            if level_of[s.out] == lv:
                place(s.out)
    for name, (b, o, _) in aliases.items():
        slots[name] = slots[b] + o
    return slots, level_of, end


def pack_mfma(steps: Sequence, inputs: Dict[str, int], aliases: Dict[str, tuple] = None,
              keep: Sequence[str] = (), perms: Dict[str, np.ndarray] = None, reuse: bool = True) -> PackedMfma:
    """keep: tensors that must survive to the end of the program (what the epilogue reads).
    reuse: let tensors share a place in the activation row once the earlier one is dead.
    perms: tensor name -> permutation p (position i of the stored tensor holds element p[i] of the program's
    tensor); tensors not named keep their order."""
    aliases = aliases or {}
    perms = perms or {}
    slots, level_of, end = _assign_slots(steps, inputs, aliases, keep, reuse)
    sizes: Dict[str, int] = {n: int(sz) for n, sz in inputs.items()}
    for name, (base, o, size) in aliases.items():
        sizes[name] = int(size)
    table, wblobs, bblobs, ctinfo, kidx = [], [], [], [], []
    w_off = b_off = macs = padded = dense = 0
    per_level: Dict[int, list] = {}
    for si, s in enumerate(steps):
        N, K = (int(x) for x in s.W.shape)
        if sizes[s.inp] != K:
            raise ValueError(f'step {s.out}: input {s.inp} has {sizes[s.inp]} floats, weight expects {K}')
        if s.skip and not np.array_equal(perms.get(s.skip, np.arange(N)), perms.get(s.out, np.arange(N))):
            raise ValueError(f'step {s.out}: the residual input must be stored in the same order as the output')
        lv = level_of[s.out]
        sizes[s.out] = N
        k16, nct = (K + 15) // 16, (N + 15) // 16
        W = s.W.detach().cpu().numpy()
        b = s.b.detach().cpu().numpy()
        if s.out in perms:
            W, b = W[perms[s.out]], b[perms[s.out]]
        if s.inp in perms:
            W = W[:, perms[s.inp]]
        Wp = np.zeros((nct * 16, k16 * 16), np.float32)
        Wp[:N, :K] = W
        # [ct][col16][k16][kk][j] -> [ct][k16][kk*16 + col16][j]
        frag = Wp.reshape(nct, 16, k16, 4, 4).transpose(0, 2, 3, 1, 4).reshape(nct, k16, 256)
        bp = np.zeros(nct * 16, np.float32)
        bp[:N] = b
        table += [lv - 1, N, k16, slots[s.inp], slots[s.out], slots[s.skip] if s.skip else -1, ACT[s.act], w_off, b_off,
                  nct, len(ctinfo) // 3, 0]
        for ct in range(nct):
            live = [kb for kb in range(k16) if np.any(frag[ct, kb] != 0.0)]
            ctinfo += [w_off, len(kidx), len(live)]
            kidx += live
            for kb in live:
                wblobs.append(frag[ct, kb])
            w_off += 256 * len(live)
            padded += 256 * len(live)
            per_level.setdefault(lv - 1, []).append((len(live), si, ct))
        bblobs.append(bp)
        b_off += bp.size
        macs += N * K
        dense += nct * 16 * k16 * 16
    n_levels = max(per_level) + 1 if per_level else 0
    parts = []
    for n_waves, may_split in ((8, True), (4, False)):
        wstart, wjobs = _schedule(per_level, n_levels, n_waves, may_split)
        parts.append((wstart, _descriptors(wjobs, table, ctinfo, kidx)))
    sched, off = [0, 0, 0, 0], 4
    for k, (wstart, wdesc) in enumerate(parts):
        sched[2 * k], sched[2 * k + 1] = off, off + len(wstart)
        off += len(wstart) + len(wdesc)
    for wstart, wdesc in parts:
        sched += wstart + wdesc
    # an input is read in blocks of 16 floats, i.e. up to 15 floats past its own end, where the weights
    # are zero but the values must exist: the row covers the furthest such read
    tab = np.asarray(table, np.int64).reshape(-1, 12)
    stride = max(end, int((tab[:, 3] + tab[:, 2] * 16).max()) if len(tab) else 0)
    stride = (stride + 3) // 4 * 4
    if stride % 8 == 0:
        stride += 4                      # 4 x odd: rows start 4 x (odd multiple) banks apart
    wfrag = np.concatenate(wblobs) if wblobs else np.zeros(256, np.float32)
    return PackedMfma(np.asarray(table, np.int32), n_levels, np.asarray(sched, np.int32),
                      np.ascontiguousarray(wfrag, np.float32), np.concatenate(bblobs), int(stride), slots, macs, padded,
                      np.asarray(ctinfo, np.int32), np.asarray(kidx if kidx else [0], np.int32), 0, dense)


def _schedule(per_level, n_levels, n_waves, may_split):
    """Longest-first assignment of each level's jobs (blocks, step, ct) to `n_waves` waves; with `may_split` a job
    that covers both 16-row halves of a 32-tree tile can give one half to the least loaded wave.  Cost model: 8 matrix
    instructions per block for both halves, 4 for one, plus a fixed cost per job (output stores, table reads).
    Returns (wstart, wjobs) as flat int lists."""
    cost = lambda nkb, mode: (8 if mode == 0 else 4) * nkb + 6  # noqa: E731
    per_wave = [[[] for _ in range(n_levels)] for _ in range(n_waves)]
    for lv in range(n_levels):
        load = [0] * n_waves
        for nkb, si, ct in sorted(per_level.get(lv, []), key=lambda x: (-x[0], x[1], x[2])):
            w = min(range(n_waves), key=lambda i: (load[i], i))
            job = [nkb, si, ct, 0 if may_split else 1]
            per_wave[w][lv].append(job)
            load[w] += cost(nkb, job[3])
        for _ in range(4 * n_waves if may_split else 0):
            hi = max(range(n_waves), key=lambda i: (load[i], -i))
            lo = min(range(n_waves), key=lambda i: (load[i], i))
            whole = [j for j in per_wave[hi][lv] if j[3] == 0]
            if not whole:
                break
            gap = load[hi] - load[lo]
            j = min(whole, key=lambda j: abs(cost(j[0], 1) - gap / 2))
            new_hi, new_lo = load[hi] - cost(j[0], 0) + cost(j[0], 1), load[lo] + cost(j[0], 1)
            if max(new_hi, new_lo) >= load[hi]:
                break
            j[3] = 1
            per_wave[lo][lv].append([j[0], j[1], j[2], 2])
            load[hi], load[lo] = new_hi, new_lo
    wstart, wjobs = [], []
    for w in range(n_waves):
        for lv in range(n_levels):
            wstart.append(len(wjobs) // 3)
            for nkb, si, ct, mode in per_wave[w][lv]:
                wjobs += [si, ct, mode]
        wstart.append(len(wjobs) // 3)
        if (wstart[-1] - wstart[-1 - n_levels]) > 64:
            # a wave keeps its job descriptors in one register per field (64 lanes): a network with more jobs gets an
            # EMPTY schedule for this tile size, which tells the engine to evaluate it level by level instead
            return [0] * (n_waves * (n_levels + 1)), []
    return wstart, wjobs


def _descriptors(wjobs, table, ctinfo, kidx):
    """The per-job descriptors (DESC_INTS ints each) of a schedule's (step, ct, mode) triples."""
    out = []
    for j in range(0, len(wjobs), 3):
        si, ct, mode = wjobs[j:j + 3]
        st = table[12 * si:12 * si + 12]
        w_off, kb0, nkb = ctinfo[3 * (st[10] + ct):3 * (st[10] + ct) + 3]
        blocks = list(kidx[kb0:kb0 + nkb])[:12]
        if any(b > 255 for b in blocks):
            raise ValueError('a layer with more than 4096 inputs does not fit the job descriptors')
        packed = [0, 0, 0]
        for i, b in enumerate(blocks):
            packed[i // 4] |= int(b) << (8 * (i % 4))
        packed = [x - (1 << 32) if x >= (1 << 31) else x for x in packed]
        out += [4 * w_off, nkb, mode, st[3], st[4] + 16 * ct, (st[5] + 16 * ct) if st[5] >= 0 else -1, st[6], st[8] + 16 * ct,
                min(16, st[1] - 16 * ct), kb0] + packed
    return out


def plane_major_permutation(channels: int, cells: int) -> np.ndarray:
    """p with p[cell * channels + c] = c * cells + cell: [channel][cell] tensors stored [cell][channel]."""
    return np.arange(channels * cells).reshape(channels, cells).T.reshape(-1).copy()


def pack_network_mfma(net, plane_major=None, reuse=None):
    """Recurrent program (dynamics + prediction of the parent state) of a lowered network in MFMA
    layout, with the io vector of PackedNetwork.io().
    reuse: whether dead tensors give their place in the activation row to later ones.  Default: yes for up to 16
    actions (those searches keep a tile's activations in LDS, where a short row matters), no beyond (those searches
    run the levels across workgroups in one launch, which needs every tensor in a place of its own).
    Board networks whose channel count is a multiple of 16 store every [channel][cell] feature map of the
    program cell-major, which turns the zero pattern of their lowered convolutions into whole zero blocks
    (`plane_major`, default: on when it applies); stored hidden states keep the reference's order, the kernels
    translate when they load the parent's state and store the new one."""
    H, asz = net.H, net.action_size
    steps = list(net.dyn.steps) + list(net.pred.steps)
    perms = {}
    C = int(net.norm_groups)
    can = getattr(net, 'kind', '') == 'resnet' and C % 16 == 0 and H % C == 0
    if plane_major is None:
        plane_major = can
    if plane_major and not can:
        raise ValueError('plane-major storage needs a board network whose channel count is a multiple of 16')
    if plane_major:
        p = plane_major_permutation(C, H // C)
        perms = {s.out: p for s in steps if int(s.W.shape[0]) == H}
        perms['hidden'] = p
        perms['xa'] = np.concatenate([p, H + np.arange(asz)])
    if reuse is None:
        reuse = net.A <= 16
    pm = pack_mfma(steps, {'xa': H + asz}, {'hidden': ('xa', 0, H)},
                   keep=('next_raw', 'reward_logits', 'policy_logits', 'value_logits'), perms=perms, reuse=reuse)
    pm.plane_major = int(bool(plane_major))
    io = np.asarray([pm.slots['xa'], pm.slots['xa'] + H, pm.slots['next_raw'], pm.slots['reward_logits'],
                     pm.slots['policy_logits'], pm.slots['value_logits'], H, net.A, asz,
                     2 * net.support_limit + 1, net.norm_groups, net.obs_size], np.int32)
    return pm, io


def pack_root_network_mfma(net, plane_major=None):
    """The two programs of the initial inference (network.py:86-103) of a lowered network in MFMA layout, each with an
    io vector shaped like PackedNetwork.io(): (representation program, io), (prediction program, io).  The
    representation program reads the observation at io[0] and leaves the raw hidden state at io[2]; the prediction
    program reads the scaled hidden state at io[0] (io[8] = 0: no action beside it) and leaves policy logits at io[4],
    value logits at io[5].  Board networks keep their feature maps cell-major as in pack_network_mfma."""
    H = net.H
    C = int(net.norm_groups)
    can = getattr(net, 'kind', '') == 'resnet' and C % 16 == 0 and H % C == 0
    if plane_major is None:
        plane_major = can
    if plane_major and not can:
        raise ValueError('plane-major storage needs a board network whose channel count is a multiple of 16')
    p = plane_major_permutation(C, H // C) if plane_major else None
    out = []
    for steps, inputs, keep in ((list(net.rep.steps), {'obs': net.obs_size}, ('hidden_raw',)),
                                (list(net.pred.steps), {'hidden': H}, ('policy_logits', 'value_logits'))):
        perms = {}
        if plane_major:
            perms = {s.out: p for s in steps if int(s.W.shape[0]) == H}
            if 'hidden' in inputs:
                perms['hidden'] = p
        pm = pack_mfma(steps, inputs, None, keep=keep, perms=perms)
        pm.plane_major = int(bool(plane_major))
        first = pm.slots['obs'] if 'obs' in inputs else pm.slots['hidden']
        io = np.asarray([first, first + (H if 'hidden' in inputs else net.obs_size), pm.slots.get('hidden_raw', 0), 0,
                         pm.slots.get('policy_logits', 0), pm.slots.get('value_logits', 0), H, net.A, 0,
                         2 * net.support_limit + 1, net.norm_groups, net.obs_size], np.int32)
        out.append((pm, io))
    return out[0], out[1]
