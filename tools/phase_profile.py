#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the tuned search kernel (block 0), from the -DMZ_STAMPS build.
    python -m muzero_amd.build --stamps && python tools/phase_profile.py [cartpole|tictactoe]
Read the SHARES; the stamped build's total time is not a performance number."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np  # noqa: E402

from muzero_amd import planner as pl  # noqa: E402

pl.LIB_PATH = os.environ.get('MZ_STAMPS_LIB', os.path.join(REPO, 'muzero_amd', 'lib', 'libmzplanner_hip_stamps.so'))
from helpers import build_mlp, mlp_case  # noqa: E402

NAMES = ['root: inference + prior', 'select + gather', 'root: tables', 'D1 dyn layer 1', 'D2 dyn layer 2 + barrier', 'reduce + normalise', 'reward head + barrier', 'value head + barrier', 'softmax (2 rows)', 'backup', 'finish']


def main():
    g = sys.argv[1] if len(sys.argv) > 1 else 'cartpole'
    board = g == 'tictactoe'
    net = build_mlp(mlp_case(g))
    B, S = 4096, 25 if board else 50
    kw = dict(num_simulations=S, discount=1.0 if board else 0.997, is_board_game=board, known_bounds=(-1.0, 1.0) if board else None)
    if g == 'lunar':  # the LunarLander-shaped leg of bench.py: four actions, the classic-control net, synthetic env
        kw.update(root_dirichlet_alpha=0.25, root_exploration_eps=0.25)
    if len(sys.argv) > 2 and sys.argv[2] == 'nonoise':
        kw['root_dirichlet_alpha'] = 0.0  # root prior without Dirichlet sampling: isolates its cost in the root phase
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=B, **kw), 0)
    p.load_state_dict(net.state_dict())
    p.selfplay_reset(pl.ENV_TICTACTOE if board else (pl.ENV_SYNTHETIC if g == 'lunar' else pl.ENV_CARTPOLE))
    p.selfplay_step(-1.0 if board else 1.0, 5)
    p.lib.mz_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    tot = np.zeros(16)
    n = 10
    for _ in range(n):
        p.selfplay_step(-1.0 if board else 1.0, 1)
        st = (C.c_longlong * 16)()
        p.lib.mz_debug_read_stamps(p.h, st)
        tot += np.array(st[:], dtype=np.float64)
    tot /= n
    sub = np.array(st[11:15], dtype=np.float64) / ((n + 5) * S)  # cumulative over all launches of this process
    root_noise = tot[15]
    tot[11:] = 0
    total = tot.sum() + root_noise
    cn = (C.c_longlong * 8)()
    p.lib.mz_debug_read_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    p.lib.mz_debug_read_counters(p.h, cn)
    ts = (C.c_longlong * 32)()
    p.lib.mz_debug_read_tree_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    p.lib.mz_debug_read_tree_stamps(p.h, ts)
    if ts[9]:
        d = float(ts[9])
        print(f'tree2_select (wave 0 of block 0, per call; each stamp itself ~40 cycles): prelude {ts[0] / d:.0f}; phase A {ts[2] / d:.0f} cycles in {ts[1] / d:.2f} '
              f'iterations ({ts[2] / max(ts[1], 1):.0f} each); phase B {ts[3] / d:.2f} rounds: entry+pUCT {ts[4] / max(ts[3], 1):.0f}, max+ties {ts[5] / max(ts[3], 1):.0f}, '
              f'pick+child {ts[6] / max(ts[3], 1):.0f}, bookkeeping {ts[7] / max(ts[3], 1):.0f} each; epilogue {ts[8] / d:.0f}')
    if ts[24]:
        d = float(n + 5)  # every move of this process ran the root once
        print('root inference (per move): ' + ', '.join(f'{nm} {ts[24 + i] / d:.0f}' for i, nm in enumerate(['rep0', 'rep1', 'normalise', 'pol0+val0', 'pol1+val1', 'softmax+scalars', 'root prior', '(K-split GEMM loops of wave 0, all layers)'])))
    if ts[21]:
        d = float(ts[21])
        print(f'tree2_backup (per call): expand+loads {ts[12] / d:.0f}, value chain {ts[13] / d:.0f}, update {ts[14] / d:.0f}, min-max reduce {ts[15] / d:.0f}, '
              f'bookkeeping {ts[16] / d:.0f}, pass-2 action loop {ts[17] / d:.0f}, cache+resume {ts[18] / d:.0f}')
    if cn[0]:
        print(f'tree counters (counters build; timings below are distorted): levels {cn[0]}, cache hits {cn[1]} ({100 * cn[1] / cn[0]:.1f}%), '
              f'descents {cn[2]}, mean depth {cn[0] / max(cn[2], 1):.2f}, min-max changes per descent {cn[3] / max(cn[2], 1):.3f}')
    if cn[2] and cn[4]:
        print(f'level evaluations (2-action path): {cn[4] / cn[2]:.2f} per descent; both children visited {100 * cn[5] / cn[4]:.1f}%, of all: cached choice '
              f'still stood {100 * cn[6] / cn[4]:.1f}%, one unvisited + still stood {100 * cn[7] / cn[4]:.1f}%; min-max changes per descent {cn[3] / cn[2]:.3f}')
    print(f'{g}: total stamped ticks per move (block 0): {total:.0f}  (s_memtime ticks, 100 MHz on gfx950)')
    print(f'  {"root: noise + obs":18s} {root_noise:10.0f}  {100 * root_noise / total:5.1f}%')
    for i, name in enumerate(NAMES):
        per_sim = tot[i] / (S if 1 <= i <= 9 else 1)
        print(f'  {name:26s} {tot[i]:10.0f}  {100 * tot[i] / total:5.1f}%   per-sim {per_sim:8.1f}')


if __name__ == '__main__':
    main()
