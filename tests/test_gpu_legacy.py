"""`legacy_scalar_promotion` on the GPU (include/mzplanner.h mz_config): child_U's product for searches without root noise in
the numpy-1.21 form (float64, one rounding) -- the reference's pinned numpy -- instead of the numpy-2 form (float32) that is the default.  The
reference's own searches under both forms are in tests/golden/legacy_cases.npz (oracle/gen_golden.py legacy; tests/test_oracle_legacy.py pins the
oracle to them on the CPU): scripted trees where the two forms take different paths run through the tree kernels (LDS trees: selection cache and
per-level schedule; HBM trees), and deterministic searches on seeded networks through the tuned, the shape-generic and the conv-tower paths."""
import numpy as np
import pytest

from helpers import build_conv, build_mlp, conv_case, load_golden, mlp_case

pytestmark = pytest.mark.gpu
G = load_golden('legacy_cases.npz')
N_TREES = int(G['lt_n'])


def _planner(net, num_envs, **search):
    from muzero_amd import planner as pl

    p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=num_envs, **search), 0)
    p.load_state_dict(net.state_dict())
    return p


def _tree(i):
    return {k[len(f'lt{i}_'):]: G[k] for k in G.files if k.startswith(f'lt{i}_')}


@pytest.mark.parametrize('path', ['cached', 'per_level', 'hbm'])
@pytest.mark.parametrize('i', range(N_TREES))
def test_scripted_trees_follow_the_flag_on_every_tree_implementation(i, path, monkeypatch):
    if path == 'per_level':
        monkeypatch.setenv('MZ_TREE_OLD', '1')
    if path == 'hbm':
        monkeypatch.setenv('MZ_HBM_TREE', '1')
    c = _tree(i)
    A, S = int(c['A']), int(c['sims'])
    net = build_mlp(('x', (4,), A, 16, 1, 1, 16, 1))
    kb = (float(c['kb_min']), float(c['kb_max'])) if int(c['has_bounds']) else None
    B = 3
    rep = lambda x: np.repeat(np.asarray(x)[None], B, axis=0)  # noqa: E731
    for legacy, tp, ta in ((True, c['trace_parent'], c['trace_action']), (False, c['numpy2_trace_parent'], c['numpy2_trace_action'])):
        p = _planner(net, 16, num_simulations=S, discount=float(c['discount']), is_board_game=bool(c['board']), known_bounds=kb,
                     root_dirichlet_alpha=float(c['alpha']), root_exploration_eps=float(c['eps']), pb_c_base=float(c['pb_c_base']),
                     pb_c_init=float(c['pb_c_init']), legacy_scalar_promotion=legacy)
        r = p.search_scripted(rep(c['pi0']), rep(c['values']), rep(c['rewards']), rep(c['mask']), int(c['cur_player']), int(c['opp_player']), 1.0, True,
                              noise=rep(np.zeros(A)), u_tie=rep(c['u_tie']), u_final=float(c['u_final']))
        for b in range(B):
            np.testing.assert_array_equal(r['trace_parent'][b], tp)
            np.testing.assert_array_equal(r['trace_action'][b], ta)
        if legacy:
            np.testing.assert_array_equal(r['visits'][0], c['visits'])
            np.testing.assert_array_equal(r['pi'][0], c['out_pi'])
            assert r['action'][0] == int(c['out_action']) and r['root_value'][0] == float(c['out_root_value'])
        p.close()


@pytest.mark.parametrize('g,kind,cname,generic', [('cartpole', 'mlp', 'cartpole', False), ('cartpole', 'mlp', 'cartpole', True), ('tictactoe', 'mlp', 'tictactoe', False),
                                                  ('board3', 'conv', 'board3', False)])
def test_network_searches_match_the_reference_under_numpy_121_promotion(g, kind, cname, generic, monkeypatch):
    """Deterministic searches (the evaluators' mode, pipeline.py:374,468) with the flag: visit counts, policy, action exact, root value 1e-4."""
    if generic:
        monkeypatch.setenv('MZ_FORCE_GENERIC', '1')
    net = build_mlp(mlp_case(cname)) if kind == 'mlp' else build_conv(conv_case(cname))
    n = int(G[f'{g}_n'])
    p = _planner(net, n, num_simulations=int(G[f'{g}_sims']), discount=float(G[f'{g}_discount']), is_board_game=bool(G[f'{g}_board']),
                 known_bounds=(float(G[f'{g}_kb_min']), float(G[f'{g}_kb_max'])) if int(G[f'{g}_has_bounds']) else None,
                 root_dirichlet_alpha=float(G[f'{g}_alpha']), root_exploration_eps=float(G[f'{g}_eps']), legacy_scalar_promotion=True)
    st = lambda k: np.stack([G[f'{g}_{j}_{k}'] for j in range(n)])  # noqa: E731
    cur, opp = st('cur_player'), st('opp_player')
    r = p.search(st('obs').astype(np.float32), st('mask').astype(bool), cur, opp, 1.0, True, noise=np.zeros((n, net.num_actions)), u_tie=st('u_tie'),
                 u_final=st('u_final'))
    np.testing.assert_array_equal(r['visits'], st('visits'))
    np.testing.assert_array_equal(r['pi'], st('out_pi'))
    np.testing.assert_array_equal(r['action'], st('out_action'))
    rv = st('out_root_value')
    assert np.all(np.abs(r['root_value'] - rv) <= 1e-4 * np.maximum(1.0, np.abs(rv)))
    assert 'MZ_FORCE_GENERIC=%d' % int(generic) in p.describe()
    p.close()
