"""Soak parity (GPU): full-size lock-step searches with fresh random roots and draws per seed, EVERY env compared bit for bit
with an independent oracle search.  The other tests pin a few hundred searches per configuration; the rare-event paths (the
two-action lead test near its slack, float32 rounding midpoints of norm_q, ties) want millions of simulations:
    MZ_SOAK_SEEDS=40 python -m pytest tests/test_gpu_soak.py -m gpu -q        (default: 2 seeds per configuration)"""
import os

import numpy as np
import pytest

from helpers import build_mlp, mlp_case
from test_oracle_nets import _oracle_net

pytestmark = pytest.mark.gpu
SEEDS = int(os.environ.get('MZ_SOAK_SEEDS', '2'))


@pytest.mark.parametrize('g', ['cartpole', 'tictactoe', 'lunar'])
def test_every_env_of_full_size_batches_equals_the_oracle(oracle, g):
    from muzero_amd import planner as pl

    case = mlp_case(g)
    net = build_mlp(case)
    onet = _oracle_net(oracle, net, 'mlp')
    A = case[2]
    board = g == 'tictactoe'
    B, S = 4096, 25 if board else 50
    kw = dict(num_simulations=S, discount=1.0 if board else 0.997, is_board_game=board, known_bounds=(-1.0, 1.0) if board else None,
              root_dirichlet_alpha=0.25, root_exploration_eps=0.25)
    ocfg = oracle.make_config(A, S, kw['discount'], board, kw['known_bounds'], 0.25, 0.25)
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=B, **kw), 0)
    p.load_state_dict(net.state_dict())
    bad = []
    for seed in range(SEEDS):
        rs = np.random.RandomState(1000 + seed)
        scale = rs.choice([0.3, 1.0, 3.0])
        obs = rs.uniform(-scale, scale, size=(B,) + tuple(case[1])).astype(np.float32)
        mask = (rs.rand(B, A) < 0.8) if board else np.ones((B, A), bool)
        mask[np.arange(B), rs.randint(0, A, B)] = True
        cur = rs.randint(1, 3, B).astype(np.int32) if board else np.ones(B, np.int32)
        opp = (3 - cur).astype(np.int32) if board else np.ones(B, np.int32)
        temp = rs.choice([1.0, 0.5, 0.25, 0.1, 0.0], size=B)
        det = seed % 4 == 3
        noise = None if det else rs.dirichlet(np.full(A, 0.25), size=B)
        u_tie, u_final = rs.rand(B, 4 * S + 8), rs.rand(B)
        r = p.search(obs, mask, cur, opp, temp, det, noise=noise, u_tie=u_tie, u_final=u_final)
        o = oracle.uct_search_batch(ocfg, onet, obs, mask.astype(np.uint8), cur, opp, temp, det, noise=noise, u_tie=u_tie, u_final=u_final)
        for k in ('visits', 'action', 'root_value') + (() if det else ('pi',)):
            a, b = np.asarray(r[k]), np.asarray(o[k])
            ne = a != b
            if a.dtype.kind == 'f':
                ne &= ~(np.isnan(a) & np.isnan(b))  # (the reference's 0/0 policy when every visit went to an illegal child)
            if ne.any():
                rows = np.unique(np.argwhere(ne)[:, 0])
                bad.append((seed, k, len(rows), rows[:5].tolist()))
    assert not bad, f'{g}: (seed, field, envs, first envs) {bad}'
