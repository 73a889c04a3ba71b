"""GPU search vs oracle on a network TRAINED on the box.  Every other GPU parity test uses seeded random weights (checkpoints cannot
travel to the GPU box), whose value ranges are degenerate; trained networks are where near-ties between actions live.  This test
trains CartPole with the whole device pipeline for a few thousand steps (device self-play -> device epilogue -> HBM replay -> the
reference's loss / Adam -> planner reload), then searches states of evaluation episodes: deterministic and sampled searches must
equal the oracle's bit for bit, on the tuned 8-wave kernel and on the shape-generic one."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tests'))  # trained_parity.py: the training + comparison helpers (a checker: it lives with the tests)


def test_trained_network_searches_equal_oracle(oracle, monkeypatch):
    import torch
    import trained_parity as tp
    from test_oracle_nets import _oracle_net
    from muzero_amd import planner as pl
    from muzero_amd.games import CartPoleEnv

    cfg, net = tp.train(2500, envs=128, seed=3)
    net.eval()
    onet = _oracle_net(oracle, net, 'mlp')
    S = cfg.num_simulations
    ocfg = oracle.make_config(2, S, cfg.discount, False, None, cfg.root_dirichlet_alpha, cfg.root_exploration_eps)
    # states the trained policy actually visits: roll host episodes with the oracle's deterministic search
    states = []
    env = CartPoleEnv(4, seed=77)
    obs, done = env.reset(), False
    while not done and len(states) < 192:
        states.append(obs.copy())
        r = oracle.uct_search_batch(ocfg, onet, obs[None].astype(np.float32), np.ones((1, 2), np.uint8), 1, 1, 0.0, True, noise=None,
                                    u_tie=np.full((1, 4 * S + 8), 0.5), u_final=np.full(1, 0.5))
        obs, _, done, _ = env.step(int(r['action'][0]))
    assert len(states) >= 60, 'a 2500-step CartPole agent balances for a while'
    B = len(states)
    obs = np.stack(states).astype(np.float32)
    mask = np.ones((B, 2), bool)
    rs = np.random.RandomState(3)
    for generic in (False, True):
        if generic:
            monkeypatch.setenv('MZ_FORCE_GENERIC', '1')
        p = pl.Planner(pl.make_mz_config(net.planner_spec(), cfg, num_envs=B, seed=5), 0)
        p.load_state_dict(net.state_dict())
        for det in (True, False):
            rng = dict(noise=None if det else rs.dirichlet(np.full(2, cfg.root_dirichlet_alpha), size=B), u_tie=rs.rand(B, 4 * S + 8), u_final=rs.rand(B))
            r = p.search(obs, mask, 1, 1, 0.0 if det else 1.0, det, **rng)
            o = oracle.uct_search_batch(ocfg, onet, obs, mask.astype(np.uint8), 1, 1, 0.0 if det else 1.0, det, **rng)
            for k in ('visits', 'pi', 'action', 'root_value'):
                np.testing.assert_array_equal(r[k], o[k], err_msg=f'generic={generic} det={det}: {k}')
            assert np.ptp(o['root_value']) > 0.5  # trained values, not the constant of a random net
        p.close()
        monkeypatch.delenv('MZ_FORCE_GENERIC', raising=False)
