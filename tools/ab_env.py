#!/usr/bin/env python3
"""Diagnostic A/B: the C2 / C3 self-play move under different environment switches of ONE library build.
    python tools/ab_env.py "MZ_RS=0" "MZ_RS=1" "MZ_RS=0 MZ_HWX=1" ...      (each argument: space-separated VAR=value settings)"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
from muzero_amd import planner as pl
if os.environ.get('MZ_LIB'): pl.LIB_PATH = os.environ['MZ_LIB']
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
    p.selfplay_step(T, 150)
    best = 1e9; bw = 1e9
    for rep in range(3):
        p.synchronize(); t0 = time.perf_counter()
        p.profile_begin(); p.selfplay_step(T, 200); prof = p.profile_end()
        p.synchronize(); w = (time.perf_counter() - t0) / 200
        best = min(best, prof['search_kernel_ms'] / prof['search_kernel_launches']); bw = min(bw, w)
    print(f'  {g}: {best * 1e3:.1f} us per search launch, {bw * 1e6:.1f} us per move (wall)')
    p.close()
''' % (REPO, REPO)

for setting in sys.argv[1:]:
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split('=', 1)
        env[k] = v
    print(setting or '(default)', flush=True)
    subprocess.run([sys.executable, '-c', CHILD], env=env, check=False, timeout=300)
