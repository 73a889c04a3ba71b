"""GPU parity (run with -m gpu on an MI355X): the HIP planner, called through the C ABI, against
  (1) the CPU oracle on the same seeded inputs -- bit-exact (float32 network outputs, float64 policy / root value), and
  (2) the committed golden fixtures recorded from the reference implementation.
The planner computes every dot product as the same k-ordered float32 fmaf chain as the oracle (v_mfma_f32_16x16x4_f32),
uses the same exp polynomial, IEEE sqrt/div and float64 tree arithmetic, so equality is exact, not approximate."""
import numpy as np
import pytest

from helpers import MLP_CASES, build_mlp, load_golden, mlp_case
from test_oracle_nets import _oracle_net

pytestmark = pytest.mark.gpu

TREE = load_golden('tree_cases.npz')
SEARCH = load_golden('search_cases.npz')
NETS = load_golden('net_cases.npz')
PLAY = load_golden('selfplay_cases.npz')


def _planner(net, num_envs, **search):
    from muzero_amd import planner as pl

    p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=num_envs, **search), 0)
    p.load_state_dict(net.state_dict())
    return p


# ----------------------------------------------------------------------------------------------- inference
@pytest.mark.parametrize('case', MLP_CASES, ids=[c[0] for c in MLP_CASES])
def test_inference_bit_exact_vs_oracle(oracle, case):
    net = build_mlp(case)
    onet = _oracle_net(oracle, net, 'mlp')
    p = _planner(net, 64)
    rs = np.random.RandomState(5)
    B = 37  # ragged: not a multiple of the 16-env tile
    obs = rs.uniform(-1, 1, size=(B,) + tuple(case[1])).astype(np.float32)
    hidden, pi, value = p.initial_inference(obs)
    actions = rs.randint(0, case[2], size=B).astype(np.int32)
    h2, reward, pi2, value2 = p.recurrent_inference(hidden, actions)
    for b in range(B):
        oh, _, opi, ov = onet.initial_inference(obs[b])
        np.testing.assert_array_equal(hidden[b], oh)
        np.testing.assert_array_equal(pi[b], opi)
        assert value[b] == np.float32(ov)
        oh2, orw, opi2, ov2 = onet.recurrent_inference(oh, int(actions[b]))
        np.testing.assert_array_equal(h2[b], oh2)
        assert reward[b] == np.float32(orw) and value2[b] == np.float32(ov2)
        np.testing.assert_array_equal(pi2[b], opi2)


def test_inference_matches_reference_fixture():
    """initial/recurrent inference vs outputs recorded from the reference network (tolerances of tests/test_oracle_nets.py)."""
    for case in MLP_CASES:
        net = build_mlp(case)
        p = _planner(net, 16)
        for j in range(3):
            pre = f'mlp_{case[0]}_{j}'
            hidden, pi, value = p.initial_inference(NETS[f'{pre}_obs'][None])
            np.testing.assert_allclose(hidden[0], NETS[f'{pre}_init_hidden'], rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(pi[0], NETS[f'{pre}_init_pi'], rtol=2e-5, atol=1e-7)
            np.testing.assert_allclose(value[0], NETS[f'{pre}_init_value'], rtol=2e-4, atol=2e-4)
            acts = NETS[f'{pre}_actions']
            hin = np.concatenate([NETS[f'{pre}_init_hidden'][None], NETS[f'{pre}_rec_hidden'][:-1]])
            h, r, pi2, v = p.recurrent_inference(hin, acts)
            np.testing.assert_allclose(h, NETS[f'{pre}_rec_hidden'], rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(r, NETS[f'{pre}_rec_reward'], rtol=2e-4, atol=2e-4)
            np.testing.assert_allclose(v, NETS[f'{pre}_rec_value'], rtol=2e-4, atol=2e-4)
            np.testing.assert_allclose(pi2, NETS[f'{pre}_rec_pi'], rtol=2e-5, atol=1e-7)


# ----------------------------------------------------------------------------------------------- tree only
def _tree_case(i):
    return {k[len(f'c{i}_'):]: TREE[k] for k in TREE.files if k.startswith(f'c{i}_')}


TREE_IDS = [i for i in range(int(TREE['num_cases'])) if int(TREE[f'c{i}_A']) <= 64]


@pytest.mark.parametrize('tree_old', ['0', '1'], ids=['cached', 'per_level'])
@pytest.mark.parametrize('i', TREE_IDS)
def test_tree_kernels_bit_exact_vs_reference(i, tree_old, monkeypatch):
    """Scripted network outputs injected: select / expand / backup / min-max / play policy of the HIP kernel against the
    REFERENCE's own tree (golden fixture), every simulation's (parent, action) included."""
    monkeypatch.setenv('MZ_TREE_OLD', tree_old)  # both tree schedules (selection cache / per-level evaluation)
    c = _tree_case(i)
    A, S = int(c['A']), int(c['sims'])
    net = build_mlp(('x', (4,), A, 16, 1, 1, 16, 1))
    kb = (float(c['kb_min']), float(c['kb_max'])) if int(c['has_bounds']) else None
    p = _planner(net, 16, num_simulations=S, discount=float(c['discount']), is_board_game=bool(c['board']), known_bounds=kb,
                 root_dirichlet_alpha=float(c['alpha']), root_exploration_eps=float(c['eps']), pb_c_base=float(c['pb_c_base']),
                 pb_c_init=float(c['pb_c_init']))
    # replicate the case in 3 env slots (different tile lanes) to catch cross-env interference
    B = 3
    rep = lambda x: np.repeat(np.asarray(x)[None], B, axis=0)  # noqa: E731
    r = p.search_scripted(rep(c['pi0']), rep(c['values']), rep(c['rewards']), rep(c['mask']), int(c['cur_player']), int(c['opp_player']),
                          float(c['temperature']), bool(c['deterministic']), noise=rep(c['noise']) if int(c['has_noise']) else rep(np.zeros(A)),
                          u_tie=rep(c['u_tie']), u_final=float(c['u_final']))
    for b in range(B):
        np.testing.assert_array_equal(r['trace_parent'][b], c['trace_parent'])
        np.testing.assert_array_equal(r['trace_action'][b], c['trace_action'])
        np.testing.assert_array_equal(r['visits'][b], c['visits'])
        T = float(c['temperature'])
        ex = max(1.0, min(5.0, 1.0 / T)) if T > 0 else 1.0
        if ex == int(ex):  # every shipped schedule: exponent 1, 2, 4 or 5 -> exact
            np.testing.assert_array_equal(r['pi'][b], c['out_pi'])
        else:  # non-integer exponent: device pow vs numpy pow may differ in the last bit
            np.testing.assert_allclose(r['pi'][b], c['out_pi'], rtol=1e-15, atol=0)
        assert r['action'][b] == int(c['out_action'])
        assert r['root_value'][b] == float(c['out_root_value'])


# ----------------------------------------------------------------------------------------------- full search
def _cfg_kwargs(G, g):
    return dict(
        num_simulations=int(G[f'{g}_sims']), discount=float(G[f'{g}_discount']), is_board_game=bool(G[f'{g}_board']),
        known_bounds=(float(G[f'{g}_kb_min']), float(G[f'{g}_kb_max'])) if int(G[f'{g}_has_bounds']) else None,
        root_dirichlet_alpha=float(G[f'{g}_alpha']), root_exploration_eps=float(G[f'{g}_eps']),
    )


@pytest.mark.parametrize('g', ['cartpole', 'tictactoe', 'lunar'])
def test_literal_numpy_seeds_reproduce_the_reference_search(g):
    """VERDICT r3 missing #4 / SURVEY appendix C: `np.random.seed(s); uct_search(..., rng='numpy')` is the reference's protocol word for word
    -- global MT19937 stream, Dirichlet noise, tie-breaks and the final sample drawn where muzero/mcts.py:302-407 draws them -- with the
    network evaluated by the HIP inference kernels.  Against the reference's own searches from the same seeds (full-size seeded nets): action and
    policy exact, root value 1e-4, and the generator left in the SAME state (the next uniform equals the reference's)."""
    import types

    import torch
    from muzero_amd import mcts

    G = SEARCH
    net = build_mlp(mlp_case(g))
    kw = _cfg_kwargs(G, g)
    kb = mcts.KnownBounds(*kw['known_bounds']) if kw['known_bounds'] else None
    cfg = types.SimpleNamespace(num_simulations=kw['num_simulations'], discount=kw['discount'], is_board_game=kw['is_board_game'], known_bounds=kb,
                                root_dirichlet_alpha=kw['root_dirichlet_alpha'], root_exploration_eps=kw['root_exploration_eps'],
                                pb_c_base=float(G[f'{g}_pb_c_base']), pb_c_init=float(G[f'{g}_pb_c_init']))
    dev = torch.device('cuda', 0)
    for j in range(int(G[f'{g}_n'])):
        pre = f'{g}_{j}'
        np.random.seed(int(G[f'{pre}_seed']))
        a, pi, root = mcts.uct_search(G[f'{pre}_obs'], net, dev, cfg, float(G[f'{pre}_temperature']), G[f'{pre}_mask'].astype(bool), int(G[f'{pre}_cur_player']),
                                      int(G[f'{pre}_opp_player']), deterministic=bool(G[f'{pre}_deterministic']), rng='numpy')
        after = float(np.random.random_sample())
        assert a == int(G[f'{pre}_out_action']), pre
        np.testing.assert_array_equal(pi, G[f'{pre}_out_pi'])
        rv = float(G[f'{pre}_out_root_value'])
        assert abs(root - rv) <= 1e-4 * max(1.0, abs(rv))
        assert after == float(G[f'{pre}_next_uniform']), pre


@pytest.mark.parametrize('g', ['cartpole', 'tictactoe', 'lunar'])
def test_search_matches_reference_fixture_and_oracle(oracle, g):
    G = SEARCH
    net = build_mlp(mlp_case(g))
    onet = _oracle_net(oracle, net, 'mlp')
    kw = _cfg_kwargs(G, g)
    A = net.num_actions
    ocfg = oracle.make_config(A, kw['num_simulations'], kw['discount'], kw['is_board_game'], kw['known_bounds'], kw['root_dirichlet_alpha'],
                              kw['root_exploration_eps'])
    p = _planner(net, 16, **kw)
    for j in range(int(G[f'{g}_n'])):
        pre = f'{g}_{j}'
        det = bool(G[f'{pre}_deterministic'])
        noise = G[f'{pre}_noise'] if int(G[f'{pre}_has_noise']) else None
        r = p.search(G[f'{pre}_obs'][None], G[f'{pre}_mask'][None], int(G[f'{pre}_cur_player']), int(G[f'{pre}_opp_player']),
                     float(G[f'{pre}_temperature']), det, noise=None if noise is None else noise[None], u_tie=G[f'{pre}_u_tie'][None],
                     u_final=float(G[f'{pre}_u_final']))
        # reference fixture: visit counts, policy, action exact; root value to 1e-4 (float32 summation order of torch)
        np.testing.assert_array_equal(r['visits'][0], G[f'{pre}_visits'])
        np.testing.assert_array_equal(r['pi'][0], G[f'{pre}_out_pi'])
        assert r['action'][0] == int(G[f'{pre}_out_action'])
        rv = float(G[f'{pre}_out_root_value'])
        assert abs(r['root_value'][0] - rv) <= 1e-4 * max(1.0, abs(rv))
        # oracle: everything bit-exact
        o = oracle.uct_search(ocfg, onet, G[f'{pre}_obs'], G[f'{pre}_mask'], int(G[f'{pre}_cur_player']), int(G[f'{pre}_opp_player']),
                              float(G[f'{pre}_temperature']), det, noise=noise, u_tie=G[f'{pre}_u_tie'], u_final=float(G[f'{pre}_u_final']))
        assert r['root_value'][0] == o['root_value']
        np.testing.assert_array_equal(r['pi'][0], o['pi'])
        np.testing.assert_array_equal(r['visits'][0], o['visits'])


@pytest.mark.parametrize('g,B', [('cartpole', 200), ('tictactoe', 200), ('lunar', 48), ('tiny', 33), ('odd', 20)])
def test_batched_search_bit_exact_vs_oracle(oracle, g, B):
    """Many envs in lock-step with random injected draws: every env equals an independent oracle search."""
    case = mlp_case(g)
    net = build_mlp(case)
    onet = _oracle_net(oracle, net, 'mlp')
    A = case[2]
    board = g == 'tictactoe'
    S = 25 if board else (50 if g in ('cartpole', 'lunar') else 17)
    kw = dict(num_simulations=S, discount=1.0 if board else 0.997, is_board_game=board, known_bounds=(-1.0, 1.0) if board else None,
              root_dirichlet_alpha=0.25, root_exploration_eps=0.25)
    ocfg = oracle.make_config(A, S, kw['discount'], board, kw['known_bounds'], 0.25, 0.25)
    p = _planner(net, B, **kw)
    rs = np.random.RandomState(123)
    obs = rs.uniform(-1, 1, size=(B,) + tuple(case[1])).astype(np.float32)
    mask = (rs.rand(B, A) < 0.8) if board else np.ones((B, A), bool)
    mask[np.arange(B), rs.randint(0, A, B)] = True
    cur = rs.randint(1, 3, B).astype(np.int32) if board else np.ones(B, np.int32)
    opp = (3 - cur).astype(np.int32) if board else np.ones(B, np.int32)
    temp = rs.choice([1.0, 0.5, 0.25, 0.1, 0.0], size=B)
    noise = rs.dirichlet(np.full(A, 0.25), size=B)
    u_tie = rs.rand(B, 4 * S + 8)
    u_final = rs.rand(B)
    r = p.search(obs, mask, cur, opp, temp, False, noise=noise, u_tie=u_tie, u_final=u_final)
    o = oracle.uct_search_batch(ocfg, onet, obs, mask.astype(np.uint8), cur, opp, temp, False, noise=noise, u_tie=u_tie, u_final=u_final)
    np.testing.assert_array_equal(r['visits'], o['visits'])
    np.testing.assert_array_equal(r['pi'], o['pi'])
    np.testing.assert_array_equal(r['action'], o['action'])
    np.testing.assert_array_equal(r['root_value'], o['root_value'])
    # deterministic mode (float32 prior path, no noise, argmax play)
    rd = p.search(obs, mask, cur, opp, temp, True, noise=None, u_tie=u_tie, u_final=u_final)
    od = oracle.uct_search_batch(ocfg, onet, obs, mask.astype(np.uint8), cur, opp, temp, True, noise=None, u_tie=u_tie, u_final=u_final)
    np.testing.assert_array_equal(rd['visits'], od['visits'])
    np.testing.assert_array_equal(rd['action'], od['action'])
    np.testing.assert_array_equal(rd['root_value'], od['root_value'])


@pytest.mark.parametrize('ishape,A,P,vs,rs_,board', [
    ((4, 5), 4, 256, 1, 1, False),      # MSE heads on a tuned shape that is not the ten-action build: shape-generic kernel (scalar_head_tile)
    ((4, 5), 2, 512, 10, 1, False),     # one categorical, one MSE head (same tile count: a tuned shape)
    ((3, 3, 3), 10, 256, 1, 1, False),  # the ten-action build outside TicTacToe's settings: single player, no known bounds
    ((3, 3, 3), 10, 256, 1, 1, True),
], ids=['mse-a4-p256', 'mixed-a2-p512', 'mse-a10-single', 'mse-a10-board'])
def test_mse_heads_on_tuned_shapes_bit_exact_vs_oracle(oracle, ishape, A, P, vs, rs_, board):
    """The one-neuron second layer of an MSE head has its own summation order (oracle: linear_mlp split == 2; kernels:
    scalar_head_tile / the ten-action tuned kernel's register chains): every kernel a tuned shape can be routed to agrees with the oracle."""
    case = ('mse', ishape, A, P, vs, rs_, 64, 31)
    net = build_mlp(case)
    onet = _oracle_net(oracle, net, 'mlp')
    S, B = 25, 80
    kw = dict(num_simulations=S, discount=1.0 if board else 0.997, is_board_game=board, known_bounds=(-1.0, 1.0) if board else None,
              root_dirichlet_alpha=0.25, root_exploration_eps=0.25)
    ocfg = oracle.make_config(A, S, kw['discount'], board, kw['known_bounds'], 0.25, 0.25)
    p = _planner(net, B, **kw)
    rs = np.random.RandomState(77)
    obs = rs.uniform(-1, 1, size=(B,) + tuple(ishape)).astype(np.float32)
    mask = np.ones((B, A), bool)
    cur = rs.randint(1, 3, B).astype(np.int32) if board else np.ones(B, np.int32)
    opp = (3 - cur).astype(np.int32) if board else np.ones(B, np.int32)
    temp = rs.choice([1.0, 0.5, 0.0], size=B)
    noise = rs.dirichlet(np.full(A, 0.25), size=B)
    u_tie = rs.rand(B, 4 * S + 8)
    u_final = rs.rand(B)
    r = p.search(obs, mask, cur, opp, temp, False, noise=noise, u_tie=u_tie, u_final=u_final)
    o = oracle.uct_search_batch(ocfg, onet, obs, mask.astype(np.uint8), cur, opp, temp, False, noise=noise, u_tie=u_tie, u_final=u_final)
    np.testing.assert_array_equal(r['visits'], o['visits'])
    np.testing.assert_array_equal(r['pi'], o['pi'])
    np.testing.assert_array_equal(r['action'], o['action'])
    np.testing.assert_array_equal(r['root_value'], o['root_value'])
    p.close()


def test_generic_and_tuned_kernels_agree(monkeypatch):
    """k_search (shape-generic) and k_search_fast (benchmark shapes) are the same algorithm: identical outputs."""
    for g, S, board in (('cartpole', 50, False), ('tictactoe', 25, True)):
        case = mlp_case(g)
        net = build_mlp(case)
        A, B = case[2], 70
        kw = dict(num_simulations=S, discount=1.0 if board else 0.997, is_board_game=board, known_bounds=(-1.0, 1.0) if board else None)
        rs = np.random.RandomState(11)
        obs = rs.uniform(-1, 1, size=(B,) + tuple(case[1])).astype(np.float32)
        args = (obs, np.ones((B, A), bool), 1, 2 if board else 1, 1.0, False)
        rng = dict(noise=rs.dirichlet(np.full(A, 0.25), size=B), u_tie=rs.rand(B, 4 * S + 8), u_final=rs.rand(B))
        monkeypatch.delenv('MZ_FORCE_GENERIC', raising=False)
        fast = _planner(net, B, **kw).search(*args, **rng)
        monkeypatch.setenv('MZ_FORCE_GENERIC', '1')
        gen = _planner(net, B, **kw).search(*args, **rng)
        monkeypatch.setenv('MZ_TREE_OLD', '1')
        gen_old = _planner(net, B, **kw).search(*args, **rng)
        monkeypatch.delenv('MZ_FORCE_GENERIC', raising=False)
        fast_old = _planner(net, B, **kw).search(*args, **rng)
        monkeypatch.delenv('MZ_TREE_OLD', raising=False)
        for k in ('visits', 'pi', 'action', 'root_value'):
            np.testing.assert_array_equal(fast[k], gen[k])
            np.testing.assert_array_equal(fast[k], gen_old[k])
            np.testing.assert_array_equal(fast[k], fast_old[k])


def test_helper_wave_work_split_does_not_change_results(monkeypatch):
    """k_search_fast runs 8 waves: waves 4-7 may take the new state's normalisation (bit 0 of MZ_HWX) and the reward row's
    softmax (bit 1) off the MFMA waves.  Every split computes the same values: identical outputs for 0, 1, 2, 3 (and the default)."""
    for g, S, board in (('cartpole', 50, False), ('lunar', 50, False), ('tictactoe', 25, True)):
        case = mlp_case(g)
        net = build_mlp(case)
        A, B = case[2], 52
        kw = dict(num_simulations=S, discount=1.0 if board else 0.997, is_board_game=board, known_bounds=(-1.0, 1.0) if board else None)
        rs = np.random.RandomState(23)
        obs = rs.uniform(-1, 1, size=(B,) + tuple(case[1])).astype(np.float32)
        args = (obs, np.ones((B, A), bool), 1, 2 if board else 1, 1.0, False)
        rng = dict(noise=rs.dirichlet(np.full(A, 0.25), size=B), u_tie=rs.rand(B, 4 * S + 8), u_final=rs.rand(B))
        monkeypatch.delenv('MZ_HWX', raising=False)
        ref = _planner(net, B, **kw).search(*args, **rng)
        for split in ('0', '1', '2', '3'):
            monkeypatch.setenv('MZ_HWX', split)
            r = _planner(net, B, **kw).search(*args, **rng)
            for k in ('visits', 'pi', 'action', 'root_value'):
                np.testing.assert_array_equal(ref[k], r[k], err_msg=f'{g}: MZ_HWX={split}: {k}')
        monkeypatch.delenv('MZ_HWX', raising=False)


def test_production_rng_properties():
    """On-device Philox mode at the BASELINE size (4096 envs): size-independent properties -- visit counts sum to
    num_simulations, policy is a distribution supported on legal actions, runs are reproducible for a fixed seed."""
    case = mlp_case('cartpole')
    net = build_mlp(case)
    B, S = 4096, 50
    kw = dict(num_simulations=S, discount=0.997, root_dirichlet_alpha=0.25, root_exploration_eps=0.25)
    rs = np.random.RandomState(7)
    obs = rs.uniform(-1, 1, size=(B, 4, 5)).astype(np.float32)
    outs = []
    for _ in range(2):
        p = _planner(net, B, **kw)
        outs.append(p.search(obs, np.ones((B, 2), bool), 1, 1, 1.0, False))
    r = outs[0]
    assert (r['visits'].sum(axis=1) == S).all()
    np.testing.assert_allclose(r['pi'].sum(axis=1), 1.0, atol=1e-12)
    np.testing.assert_array_equal(r['pi'], r['visits'] / S)
    assert ((r['action'] >= 0) & (r['action'] < 2)).all()
    for k in ('visits', 'action', 'root_value'):
        np.testing.assert_array_equal(outs[0][k], outs[1][k])
    # sampled actions follow the policy: frequency of action 1 vs mean pi[1]
    assert abs(r['action'].mean() - r['pi'][:, 1].mean()) < 0.03


def test_full_size_c3_properties_and_oracle_sample(oracle):
    """BASELINE configs[2] at full size (TicTacToe MLP, 4096 envs, 25 simulations, two-player backup, bounds (-1, 1)) in
    parity mode: every 64th env is checked bit-exactly against the oracle, all envs against the size-independent properties
    (masked visit counts, policy = visits^(1/T) normalised, legal actions, a pure function of inputs and draws)."""
    case = mlp_case('tictactoe')
    net = build_mlp(case)
    onet = _oracle_net(oracle, net, 'mlp')
    B, S, A = 4096, 25, 10
    kw = dict(num_simulations=S, discount=1.0, is_board_game=True, known_bounds=(-1.0, 1.0), root_dirichlet_alpha=0.25, root_exploration_eps=0.25)
    rs = np.random.RandomState(31)
    obs = (rs.rand(B, 9, 3, 3) < 0.3).astype(np.float32)
    mask = rs.rand(B, A) < 0.7
    mask[np.arange(B), rs.randint(0, A, B)] = True
    cur = rs.randint(1, 3, B).astype(np.int32)
    opp = (3 - cur).astype(np.int32)
    noise = rs.dirichlet(np.full(A, 0.25), size=B)
    u_tie, u_final = rs.rand(B, 4 * S + 8), rs.rand(B)
    p = _planner(net, B, **kw)
    r1 = p.search(obs, mask, cur, opp, 1.0, False, noise=noise, u_tie=u_tie, u_final=u_final)
    r2 = p.search(obs, mask, cur, opp, 1.0, False, noise=noise, u_tie=u_tie, u_final=u_final)
    for k in ('visits', 'pi', 'action', 'root_value'):
        np.testing.assert_array_equal(r1[k], r2[k])
    v = r1['visits']
    ok = v.sum(1) > 0  # (all visits on illegal root children: the reference's 0/0 policy)
    assert (v[~mask] == 0).all() and (v.sum(1) <= S).all() and ok.mean() > 0.95
    np.testing.assert_array_equal(r1['pi'][ok], v[ok] / v[ok].sum(1, keepdims=True))
    assert mask[np.arange(B), r1['action']][ok].all()
    sub = np.arange(0, B, 64)
    ocfg = oracle.make_config(A, S, 1.0, True, (-1.0, 1.0), 0.25, 0.25)
    o = oracle.uct_search_batch(ocfg, onet, obs[sub], mask[sub].astype(np.uint8), cur[sub], opp[sub], np.ones(len(sub)), False, noise=noise[sub],
                                u_tie=u_tie[sub], u_final=u_final[sub])
    for k in ('visits', 'pi', 'action', 'root_value'):
        np.testing.assert_array_equal(r1[k][sub], o[k])


def test_full_size_c2_properties_and_oracle_sample(oracle):
    """BASELINE configs[1] at full size (CartPole MLP 512/64/31, 4096 envs, 50 simulations, no value bounds: the moving min-max
    pair and the two-action exact lead test of mz_tree2.h) in parity mode: every 2nd env (2048 searches, 102 400 simulations) is
    checked bit-exactly against the oracle, all envs against the size-independent properties."""
    case = mlp_case('cartpole')
    net = build_mlp(case)
    onet = _oracle_net(oracle, net, 'mlp')
    B, S, A = 4096, 50, 2
    kw = dict(num_simulations=S, discount=0.997, root_dirichlet_alpha=0.25, root_exploration_eps=0.25)
    rs = np.random.RandomState(77)
    obs = rs.uniform(-1.5, 1.5, size=(B,) + tuple(case[1])).astype(np.float32)
    mask = np.ones((B, A), bool)
    cur = np.ones(B, np.int32)
    temp = rs.choice([1.0, 0.5, 0.25], size=B)
    noise = rs.dirichlet(np.full(A, 0.25), size=B)
    u_tie, u_final = rs.rand(B, 4 * S + 8), rs.rand(B)
    p = _planner(net, B, **kw)
    r1 = p.search(obs, mask, cur, cur, temp, False, noise=noise, u_tie=u_tie, u_final=u_final)
    r2 = p.search(obs, mask, cur, cur, temp, False, noise=noise, u_tie=u_tie, u_final=u_final)
    for k in ('visits', 'pi', 'action', 'root_value'):
        np.testing.assert_array_equal(r1[k], r2[k])
    v = r1['visits']
    assert (v.sum(1) == S).all()
    sub = np.arange(0, B, 2)
    ocfg = oracle.make_config(A, S, 0.997, False, None, 0.25, 0.25)
    o = oracle.uct_search_batch(ocfg, onet, obs[sub], mask[sub].astype(np.uint8), cur[sub], cur[sub], temp[sub], False, noise=noise[sub],
                                u_tie=u_tie[sub], u_final=u_final[sub])
    for k in ('visits', 'pi', 'action', 'root_value'):
        np.testing.assert_array_equal(r1[k][sub], o[k])


def test_selfplay_episode_matches_reference_fixture(oracle):
    """pipeline.py:41-167 on TicTacToe replayed through uct_search (B=1 API) with the recorded draws."""
    from muzero_amd import mcts
    from muzero_amd.config import make_tictactoe_config
    import torch

    net = build_mlp(mlp_case('tictactoe'))
    cfg = make_tictactoe_config(use_tensorboard=False)
    dev = torch.device('cuda', 0)
    for ep in range(int(PLAY['n_episodes'])):
        for t in range(int(PLAY[f'ep{ep}_n_moves'])):
            a, pi, v = mcts.uct_search(
                PLAY[f'ep{ep}_search_obs'][t], net, dev, cfg, float(PLAY[f'ep{ep}_search_T'][t]), PLAY[f'ep{ep}_search_mask'][t].astype(bool),
                int(PLAY[f'ep{ep}_search_cur'][t]), int(PLAY[f'ep{ep}_search_opp'][t]),
                rng=dict(noise=PLAY[f'ep{ep}_search_noise'][t][None], u_tie=PLAY[f'ep{ep}_search_u_tie'][t][None],
                         u_final=float(PLAY[f'ep{ep}_search_u_final'][t])),
            )
            assert a == int(PLAY[f'ep{ep}_search_action'][t])
            np.testing.assert_array_equal(pi, PLAY[f'ep{ep}_search_pi'][t])
            assert abs(v - float(PLAY[f'ep{ep}_search_root'][t])) <= 1e-4


def test_network_api_runs_on_planner():
    """MuZeroNet.initial_inference / recurrent_inference (network.py:62-111) are served by the HIP engine."""
    import torch

    net = build_mlp(mlp_case('cartpole'))
    x = torch.from_numpy(NETS['mlp_cartpole_0_obs'])[None].to('cuda')
    out = net.initial_inference(x)
    np.testing.assert_allclose(out.hidden_state, NETS['mlp_cartpole_0_init_hidden'], rtol=2e-5, atol=2e-6)
    assert out.reward == 0.0 and isinstance(out.value, float)
    out2 = net.recurrent_inference(torch.from_numpy(out.hidden_state)[None].to('cuda'), torch.tensor([[1]], device='cuda'))
    assert out2.hidden_state.shape == (64,) and out2.pi_probs.shape == (2,)
