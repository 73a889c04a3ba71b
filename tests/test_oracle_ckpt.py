"""SURVEY 8c G3 on the SHIPPED TRAINED CHECKPOINTS (/root/reference/saved_checkpoints/*, loaded the way the reference's
evaluators load them, pipeline.py:810-817): the reference's `uct_search` (mcts.py:302-407) ran on them in the build
container with its numpy draws recorded (oracle/gen_golden.py ckpt -> tests/golden/ckpt_cases.npz: inputs, draws, outputs --
no weights); here the same roots and draws go through the oracle with the checkpoint loaded into `muzero_amd.network`.
Trained networks are where near-ties live (random weights give degenerate value ranges), so this is the sharpest check that
the oracle's arithmetic order does not flip a selection: visit counts, policy and action EXACT, root value 1e-4.

Container-only: the weights cannot travel, so the test is skipped where /root/reference is absent (the GPU box)."""
import os

import numpy as np
import pytest
import torch

from helpers import load_golden
from test_oracle_nets import _oracle_net

CKPT_DIR = '/root/reference/saved_checkpoints'
pytestmark = pytest.mark.skipif(not os.path.isdir(CKPT_DIR), reason='the shipped checkpoints live in /root/reference (build container only)')

CASES = [  # fixture group, checkpoint file, MuZeroMLPNet arguments (input shape, actions, planes, value / reward support, hidden)
    ('cartpole', 'CartPole-v1_train_steps_44800', ((4, 5), 2, 512, 31, 31, 64)),
    ('lunar', 'LunarLander-v2_train_steps_58400', ((4, 9), 4, 512, 31, 31, 64)),
    ('tictactoe', 'TicTacToe_train_steps_35000', ((9, 3, 3), 10, 256, 1, 1, 64)),
]


@pytest.mark.parametrize('g,fname,net_args', CASES, ids=[c[0] for c in CASES])
def test_search_on_shipped_checkpoint_matches_reference(oracle, g, fname, net_args):
    from muzero_amd import network

    G = load_golden('ckpt_cases.npz')
    net = network.MuZeroMLPNet(*net_args)
    ck = torch.load(os.path.join(CKPT_DIR, fname), map_location='cpu', weights_only=False)
    res = net.load_state_dict(ck['network'])  # strict: the checkpoint layout is drop-in (same keys, same shapes)
    assert not res.missing_keys and not res.unexpected_keys
    net.eval()
    onet = _oracle_net(oracle, net, 'mlp')
    cfg = oracle.make_config(
        net_args[1], int(G[f'{g}_sims']), float(G[f'{g}_discount']), bool(G[f'{g}_board']),
        (float(G[f'{g}_kb_min']), float(G[f'{g}_kb_max'])) if int(G[f'{g}_has_bounds']) else None, float(G[f'{g}_alpha']),
        float(G[f'{g}_eps']), float(G[f'{g}_pb_c_base']), float(G[f'{g}_pb_c_init']),
    )
    n = int(G[f'{g}_n'])
    assert n == 32
    spread = []
    for j in range(n):
        p = f'{g}_{j}'
        r = oracle.uct_search(
            cfg, onet, G[f'{p}_obs'], G[f'{p}_mask'], int(G[f'{p}_cur_player']), int(G[f'{p}_opp_player']), float(G[f'{p}_temperature']),
            bool(G[f'{p}_deterministic']), noise=G[f'{p}_noise'] if int(G[f'{p}_has_noise']) else None, u_tie=G[f'{p}_u_tie'],
            u_final=float(G[f'{p}_u_final']),
        )
        np.testing.assert_array_equal(r['visits'], G[f'{p}_visits'], err_msg=f'{p}: visit counts')
        np.testing.assert_array_equal(r['pi'], G[f'{p}_out_pi'], err_msg=f'{p}: policy')
        assert r['action'] == int(G[f'{p}_out_action']), p
        rv = float(G[f'{p}_out_root_value'])
        assert abs(r['root_value'] - rv) <= 1e-4 * max(1.0, abs(rv)), (p, r['root_value'], rv)
        spread.append(rv)
    # trained value ranges, not the degenerate ones of random weights: CartPole / LunarLander roots differ across states
    assert np.ptp(spread) > (0.05 if g == 'tictactoe' else 1.0)
