#!/usr/bin/env python3
"""Diagnostic A/B: time the C2 / C3 self-play move with alternative builds of the planner library.
    python tools/ab_bench.py lib_a.so lib_b.so ...      (variants built with MZ_EXTRA_FLAGS / MZ_STAMPS_OUT)"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
from muzero_amd import planner as pl
pl.LIB_PATH = sys.argv[1]
from helpers import build_mlp, mlp_case
for g in ('cartpole', 'tictactoe'):
    board = g == 'tictactoe'
    net = build_mlp(mlp_case(g))
    S = 25 if board else 50
    kw = dict(num_simulations=S, discount=1.0 if board else 0.997, is_board_game=board, known_bounds=(-1.0, 1.0) if board else None)
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=4096, seed=1000, **kw), 0)
    p.load_state_dict(net.state_dict())
    p.selfplay_reset(pl.ENV_TICTACTOE if board else pl.ENV_CARTPOLE)
    T = -1.0 if board else 1.0
    p.selfplay_step(T, 10)
    best = 1e9
    for rep in range(3):
        p.profile_begin(); p.selfplay_step(T, 100); prof = p.profile_end()
        best = min(best, prof['search_kernel_ms'] / prof['search_kernel_launches'])
    print(f'  {g}: {best * 1e3:.1f} us per search launch')
    p.close()
net = build_mlp(mlp_case('lunar'))  # the LunarLander-shaped leg of bench.py (four actions, synthetic env)
p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=4096, seed=2000, num_simulations=50, discount=0.997, root_dirichlet_alpha=0.25,
                                 root_exploration_eps=0.25), 0)
p.load_state_dict(net.state_dict())
p.selfplay_reset(pl.ENV_SYNTHETIC)
p.selfplay_step(1.0, 30)
best = 1e9
for rep in range(3):
    p.profile_begin(); p.selfplay_step(1.0, 100); prof = p.profile_end()
    best = min(best, prof['search_kernel_ms'] / prof['search_kernel_launches'])
print(f'  lunar: {best * 1e3:.1f} us per search launch')
p.close()
''' % (REPO, REPO)

for lib in sys.argv[1:]:
    print(lib, flush=True)
    subprocess.run([sys.executable, '-c', CHILD, os.path.abspath(lib)], check=False)
