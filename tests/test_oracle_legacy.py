"""The numpy-1.21 form of child_U for searches without root noise (`legacy_scalar_promotion` in include/mzplanner.h).
The reference pins numpy 1.21.6 (requirements.txt:21), under which `child.prior * <python float>` (mcts.py:189-197) with an np.float32 prior is a
FLOAT64 product rounded once; numpy >= 2 -- this container, where every other fixture was recorded -- multiplies in float32.  The fixture
tests/golden/legacy_cases.npz holds the reference's deterministic searches with the 1.21 promotion emulated exactly (oracle/gen_golden.py legacy:
child priors stored as float64 scalars): scripted trees chosen so that the two forms provably take different paths, searches on seeded networks,
and searches on the shipped trained checkpoints (container-only: the weights stay in /root/reference).  Checked here on the CPU: the oracle with the
flag (exact), the oracle without it (differs where the fixture says so), and the host `mcts.Node` walker."""
import os
import types

import numpy as np
import pytest
import torch

from helpers import build_conv, build_mlp, conv_case, load_golden, mlp_case
from test_oracle_nets import _oracle_net

G = load_golden('legacy_cases.npz')
N_TREES = int(G['lt_n'])


def _tree(i):
    return {k[len(f'lt{i}_'):]: G[k] for k in G.files if k.startswith(f'lt{i}_')}


def _ocfg(oracle, c, A, legacy, pre=''):
    g = lambda k: c[pre + k]  # noqa: E731
    return oracle.make_config(A, int(g('sims')), float(g('discount')), bool(g('board')), (float(g('kb_min')), float(g('kb_max'))) if int(g('has_bounds')) else None,
                              float(g('alpha')), float(g('eps')), float(g('pb_c_base')), float(g('pb_c_init')), legacy_scalar_promotion=legacy)


@pytest.mark.parametrize('i', range(N_TREES))
def test_scripted_trees_oracle_follows_the_flag(oracle, i):
    c = _tree(i)
    A = int(c['A'])
    res = {}
    for legacy in (True, False):
        net = oracle.Net.scripted(c['pi0'], c['values'], c['rewards'])
        res[legacy] = oracle.uct_search(_ocfg(oracle, c, A, legacy), net, np.zeros(1, np.float32), c['mask'], int(c['cur_player']), int(c['opp_player']), 1.0, True,
                                        u_tie=c['u_tie'], u_final=float(c['u_final']))
    r = res[True]  # numpy 1.21's arithmetic: the reference's trace under the emulation, bit for bit
    np.testing.assert_array_equal(r['trace_parent'], c['trace_parent'])
    np.testing.assert_array_equal(r['trace_action'], c['trace_action'])
    np.testing.assert_array_equal(r['visits'], c['visits'])
    np.testing.assert_array_equal(r['pi'], c['out_pi'])
    assert r['action'] == int(c['out_action']) and r['root_value'] == float(c['out_root_value'])
    r2 = res[False]  # numpy 2's arithmetic on the same inputs: the trace the reference produced WITHOUT the emulation
    np.testing.assert_array_equal(r2['trace_parent'], c['numpy2_trace_parent'])
    np.testing.assert_array_equal(r2['trace_action'], c['numpy2_trace_action'])
    differs = not (np.array_equal(r['trace_parent'], r2['trace_parent']) and np.array_equal(r['trace_action'], r2['trace_action']))
    assert differs == bool(int(c['differs_from_numpy2']))


def test_the_fixture_holds_trees_where_the_forms_differ():
    assert sum(int(G[f'lt{i}_differs_from_numpy2']) for i in range(N_TREES)) >= 6


class _Scripted:
    def __init__(self, pi0, values, rewards):
        self.pi0, self.values, self.rewards, self.calls, self.trace = np.asarray(pi0, np.float32), values, rewards, 0, []

    def initial_inference(self, x):
        from muzero_amd.network import NetworkOutputs

        return NetworkOutputs(hidden_state=np.array([0.0], np.float32), reward=0.0, pi_probs=self.pi0.copy(), value=0.123)

    def recurrent_inference(self, hidden_state, action):
        from muzero_amd.network import NetworkOutputs

        s = self.calls
        self.calls += 1
        self.trace.append((int(hidden_state.reshape(-1)[0].item()), int(action.reshape(-1)[0].item())))
        return NetworkOutputs(hidden_state=np.array([float(s + 1)], np.float32), reward=float(np.float32(self.rewards[s])), pi_probs=self.pi0.copy(),
                              value=float(np.float32(self.values[s])))


@pytest.mark.parametrize('i', range(N_TREES))
def test_host_node_walker_follows_the_flag(i):
    """muzero_amd.mcts.Node / uct_search(rng='numpy') with `config.legacy_scalar_promotion`: the host-side tree reproduces both forms."""
    from muzero_amd import mcts

    c = _tree(i)
    kb = mcts.KnownBounds(float(c['kb_min']), float(c['kb_max'])) if int(c['has_bounds']) else None
    for legacy, tp, ta in ((True, c['trace_parent'], c['trace_action']), (False, c['numpy2_trace_parent'], c['numpy2_trace_action'])):
        cfg = types.SimpleNamespace(discount=float(c['discount']), pb_c_base=float(c['pb_c_base']), pb_c_init=float(c['pb_c_init']), is_board_game=bool(c['board']),
                                    known_bounds=kb, num_simulations=int(c['sims']), root_dirichlet_alpha=float(c['alpha']), root_exploration_eps=float(c['eps']),
                                    legacy_scalar_promotion=legacy)
        net = _Scripted(c['pi0'], c['values'], c['rewards'])
        np.random.seed(int(c['seed']) if 'seed' in c else 0)
        mcts.uct_search(np.zeros(1, np.float32), net, torch.device('cpu'), cfg, 1.0, c['mask'].astype(bool), int(c['cur_player']), int(c['opp_player']),
                        deterministic=True, rng='numpy')
        assert net.trace == list(zip(tp.tolist(), ta.tolist())), legacy


def _search_group(oracle, onet, g, A, legacy=True):
    cfg = _ocfg(oracle, G, A, legacy, pre=f'{g}_')
    for j in range(int(G[f'{g}_n'])):
        p = f'{g}_{j}'
        r = oracle.uct_search(cfg, onet, G[f'{p}_obs'], G[f'{p}_mask'], int(G[f'{p}_cur_player']), int(G[f'{p}_opp_player']), 1.0, True,
                              u_tie=G[f'{p}_u_tie'], u_final=float(G[f'{p}_u_final']))
        np.testing.assert_array_equal(r['visits'], G[f'{p}_visits'], err_msg=p)
        np.testing.assert_array_equal(r['pi'], G[f'{p}_out_pi'], err_msg=p)
        assert r['action'] == int(G[f'{p}_out_action']), p
        rv = float(G[f'{p}_out_root_value'])
        assert abs(r['root_value'] - rv) <= 1e-4 * max(1.0, abs(rv)), (p, r['root_value'], rv)


@pytest.mark.parametrize('g,kind,cname', [('cartpole', 'mlp', 'cartpole'), ('tictactoe', 'mlp', 'tictactoe'), ('board3', 'conv', 'board3')])
def test_seeded_network_searches_match_the_reference_under_numpy_121_promotion(oracle, g, kind, cname):
    net = build_mlp(mlp_case(cname)) if kind == 'mlp' else build_conv(conv_case(cname))
    _search_group(oracle, _oracle_net(oracle, net, kind), g, net.num_actions)


CKPT_DIR = '/root/reference/saved_checkpoints'
CKPTS = [('cartpole', 'CartPole-v1_train_steps_44800', ((4, 5), 2, 512, 31, 31, 64)), ('lunar', 'LunarLander-v2_train_steps_58400', ((4, 9), 4, 512, 31, 31, 64)),
         ('tictactoe', 'TicTacToe_train_steps_35000', ((9, 3, 3), 10, 256, 1, 1, 64))]


@pytest.mark.skipif(not os.path.isdir(CKPT_DIR), reason='the shipped checkpoints live in /root/reference (build container only)')
@pytest.mark.parametrize('g,fname,net_args', CKPTS, ids=[c[0] for c in CKPTS])
def test_trained_checkpoint_searches_match_the_reference_under_numpy_121_promotion(oracle, g, fname, net_args):
    """Trained nets are where near-ties live: deterministic searches on the shipped checkpoints, 16 roots each, exact."""
    from muzero_amd import network

    net = network.MuZeroMLPNet(*net_args)
    net.load_state_dict(torch.load(os.path.join(CKPT_DIR, fname), map_location='cpu', weights_only=False)['network'])
    net.eval()
    _search_group(oracle, _oracle_net(oracle, net, 'mlp'), f'ckpt_{g}', net_args[1])
