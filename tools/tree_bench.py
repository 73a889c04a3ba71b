#!/usr/bin/env python3
"""Diagnostic: the tree phases alone (select + expand + backup of every simulation, no network) through the scripted-network
hook, timed with the planner's own HIP events.  python tools/tree_bench.py [cartpole|tictactoe]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np  # noqa: E402

from muzero_amd import planner as pl  # noqa: E402
from helpers import build_mlp, mlp_case  # noqa: E402


def main():
    g = sys.argv[1] if len(sys.argv) > 1 else 'cartpole'
    board = g == 'tictactoe'
    net = build_mlp(mlp_case(g))
    B, S = 4096, 25 if board else 50
    A = 10 if board else 2
    kw = dict(num_simulations=S, discount=1.0 if board else 0.997, is_board_game=board, known_bounds=(-1.0, 1.0) if board else None)
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=B, **kw), 0)
    p.load_state_dict(net.state_dict())
    rs = np.random.RandomState(0)
    pi0 = rs.dirichlet(np.ones(A), size=B).astype(np.float32)
    if board:
        values = rs.uniform(-1, 1, size=(B, S)).astype(np.float32)
        rewards = np.zeros((B, S), np.float32)
    else:  # CartPole-like: reward ~1 per step, values that keep pushing the min-max pair outwards
        values = (rs.uniform(0, 1, size=(B, S)) * np.linspace(5, 40, S)[None, :]).astype(np.float32)
        rewards = rs.uniform(0.9, 1.1, size=(B, S)).astype(np.float32)
    ms = []
    for it in range(6):
        p.profile_begin()
        r = p.search_scripted(pi0, values, rewards, None, 1, 2 if board else 1, 1.0)
        prof = p.profile_end()
        ms.append(prof['search_kernel_ms'])
    tp = r['trace_parent']
    depth = np.zeros((B, S + 1), np.int64)
    for s in range(S):
        depth[np.arange(B), s + 1] = depth[np.arange(B), tp[:, s]] + 1
    t = float(np.median(ms[1:]))
    print(f'{g}: tree-only search kernel {t * 1e3:.1f} us per move = {t * 1e3 / S:.2f} us per simulation '
          f'(~{t * 1e-3 / S * 2.3e9:.0f} cycles at 2.3 GHz); mean leaf depth {depth[:, 1:].mean():.2f}, mean of per-tile max {depth[:, 1:].reshape(B // 16, 16, S).max(1).mean():.2f}')


if __name__ == '__main__':
    main()
