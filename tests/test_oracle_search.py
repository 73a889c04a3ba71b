"""G3 + G6: end-to-end search and a full self-play episode -- oracle vs reference outputs recorded with seeded
networks and the reference's own random draws.  Networks differ from torch only in float32 summation order, so the
visit-count vector / policy / sampled action are expected to match exactly and the root value to 1e-4 (SURVEY
Appendix A sensitivity study)."""
import numpy as np
import pytest

from helpers import build_conv, build_mlp, conv_case, load_golden, mlp_case
from test_oracle_nets import _oracle_net

G = load_golden('search_cases.npz')
P = load_golden('selfplay_cases.npz')

GROUPS = [
    ('cartpole', 'mlp', 'cartpole'), ('tictactoe', 'mlp', 'tictactoe'), ('lunar', 'mlp', 'lunar'),
    ('board3', 'conv', 'board3'), ('atari_s', 'conv', 'atari_s'),
]


def _cfg(oracle, g, A):
    return oracle.make_config(
        A, int(G[f'{g}_sims']), float(G[f'{g}_discount']), bool(G[f'{g}_board']),
        (float(G[f'{g}_kb_min']), float(G[f'{g}_kb_max'])) if int(G[f'{g}_has_bounds']) else None, float(G[f'{g}_alpha']),
        float(G[f'{g}_eps']), float(G[f'{g}_pb_c_base']), float(G[f'{g}_pb_c_init']),
    )


@pytest.mark.parametrize('g,kind,case', GROUPS, ids=[g[0] for g in GROUPS])
def test_search_matches_reference(oracle, g, kind, case):
    net = build_mlp(mlp_case(case)) if kind == 'mlp' else build_conv(conv_case(case))
    onet = _oracle_net(oracle, net, kind)
    cfg = _cfg(oracle, g, net.num_actions)
    n = int(G[f'{g}_n'])
    exact = 0
    for j in range(n):
        p = f'{g}_{j}'
        r = oracle.uct_search(
            cfg, onet, G[f'{p}_obs'], G[f'{p}_mask'], int(G[f'{p}_cur_player']), int(G[f'{p}_opp_player']), float(G[f'{p}_temperature']),
            bool(G[f'{p}_deterministic']), noise=G[f'{p}_noise'] if int(G[f'{p}_has_noise']) else None, u_tie=G[f'{p}_u_tie'],
            u_final=float(G[f'{p}_u_final']),
        )
        same = np.array_equal(r['visits'], G[f'{p}_visits'])
        exact += same
        if same:
            np.testing.assert_array_equal(r['pi'], G[f'{p}_out_pi'])
            assert r['action'] == int(G[f'{p}_out_action'])
            rv = float(G[f'{p}_out_root_value'])
            assert abs(r['root_value'] - rv) <= 1e-4 * max(1.0, abs(rv))
    assert exact == n, f'{exact}/{n} searches reproduce the reference visit counts exactly'


@pytest.mark.parametrize('ep', range(int(P['n_episodes'])))
def test_selfplay_episode_matches_reference(oracle, ep):
    """pipeline.py:41-167 on TicTacToe: env + search + MC-return targets + unroll sequences."""
    net = build_mlp(mlp_case('tictactoe'))
    onet = _oracle_net(oracle, net, 'mlp')
    cfg = oracle.make_config(10, 25, 1.0, True, (-1, 1), 0.25, 0.25)
    env = oracle.BoardEnv(3, 4, 3)
    obs = env.reset()
    n = int(P[f'ep{ep}_n_moves'])
    traj = []
    for t in range(n):
        np.testing.assert_array_equal(obs, P[f'ep{ep}_search_obs'][t])
        np.testing.assert_array_equal(env.actions_mask, P[f'ep{ep}_search_mask'][t].astype(bool))
        assert env.current_player == int(P[f'ep{ep}_search_cur'][t])
        r = oracle.uct_search(
            cfg, onet, obs, env.actions_mask, env.current_player, env.opponent_player, float(P[f'ep{ep}_search_T'][t]), False,
            noise=P[f'ep{ep}_search_noise'][t], u_tie=P[f'ep{ep}_search_u_tie'][t], u_final=float(P[f'ep{ep}_search_u_final'][t]),
        )
        np.testing.assert_array_equal(r['visits'], P[f'ep{ep}_search_visits'][t])
        np.testing.assert_array_equal(r['pi'], P[f'ep{ep}_search_pi'][t])
        assert r['action'] == int(P[f'ep{ep}_search_action'][t])
        assert abs(r['root_value'] - float(P[f'ep{ep}_search_root'][t])) <= 1e-4
        player = env.current_player
        nobs, reward, done = env.step(r['action'])
        traj.append((obs, r['action'], reward, r['pi'], r['root_value'], player))
        obs = nobs
    assert done
    observations, actions, rewards, pis, roots, players = map(list, zip(*traj))
    targets = oracle.mc_return_target(rewards, players)
    prios = np.abs(np.array(roots) - targets)
    st, ac, rw, vl, pi, pr = oracle.make_unroll_sequence(observations, actions, rewards, pis, targets, prios, 5)
    np.testing.assert_array_equal(st, P[f'ep{ep}_tr_state'])
    np.testing.assert_array_equal(ac, P[f'ep{ep}_tr_action'])
    np.testing.assert_array_equal(rw, P[f'ep{ep}_tr_reward'])
    np.testing.assert_array_equal(vl, P[f'ep{ep}_tr_value'])
    np.testing.assert_array_equal(pi, P[f'ep{ep}_tr_pi'])
    np.testing.assert_allclose(pr, P[f'ep{ep}_tr_priority'], atol=1e-4)
