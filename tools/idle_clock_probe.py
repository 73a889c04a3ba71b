#!/usr/bin/env python3
"""Diagnostic: is the slow start of a bench run the env population's reset transient or the GPU's idle clocks?  Times blocks of
moves after start-up, after 3 s of idleness and after an env reset on a busy GPU (answer on MI355X: the clocks).
    python tools/idle_clock_probe.py"""
import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
from helpers import build_mlp, mlp_case
from muzero_amd import planner as pl
net = build_mlp(mlp_case('cartpole'))
p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=4096, seed=1000, num_simulations=50, discount=0.997, root_dirichlet_alpha=0.25, root_exploration_eps=0.25), 0)
p.load_state_dict(net.state_dict())
def timed(n):
    p.synchronize(); t0 = time.perf_counter(); p.selfplay_step(1.0, n); p.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
p.selfplay_reset(pl.ENV_CARTPOLE)
print('fresh reset: moves 1-20   %.4f ms' % timed(20))
print('             moves 21-40  %.4f ms' % timed(20))
print('             moves 41-60  %.4f ms' % timed(20))
print('             moves 61-160 %.4f ms' % timed(100))
print('             next 20      %.4f ms' % timed(20))
time.sleep(3.0)
print('after 3 s idle: 20 moves  %.4f ms' % timed(20))
p.selfplay_reset(pl.ENV_CARTPOLE)
print('reset again (GPU warm): moves 1-20 %.4f ms' % timed(20))
print('                        moves 21-40 %.4f ms' % timed(20))
