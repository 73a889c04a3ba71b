"""Error behaviour and edge shapes of the C ABI on a GPU box: status codes surface as PlannerError with mz_last_error's text
(no exceptions cross the ABI, no silent fallback), single-env / single-simulation / widest-action-set planners work."""
import numpy as np
import pytest

from helpers import build_conv, build_mlp, mlp_case

pytestmark = pytest.mark.gpu


def _cfg(net, **kw):
    from muzero_amd import planner as pl

    return pl.make_mz_config(net.planner_spec(), None, **kw)


def test_state_and_argument_errors():
    from muzero_amd import planner as pl

    net = build_mlp(mlp_case('tiny'))
    p = pl.Planner(_cfg(net, num_envs=4, num_simulations=5), 0)
    obs = np.zeros((2, 3, 4), np.float32)
    with pytest.raises(pl.PlannerError, match='not committed'):
        p.search(obs, None, 1, 1, 1.0)
    with pytest.raises(pl.PlannerError, match='committed'):
        p.initial_inference(obs)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    bad = dict(sd)
    bad['represent_net.net.0.weight'] = sd['represent_net.net.0.weight'][:, :-1]
    with pytest.raises(pl.PlannerError, match='shape mismatch'):
        p.load_state_dict(bad)
    missing = {k: v for k, v in sd.items() if not k.startswith('prediction_net.value_net.2')}
    p2 = pl.Planner(_cfg(net, num_envs=4, num_simulations=5), 0)
    with pytest.raises(pl.PlannerError, match='missing parameter'):
        p2.load_state_dict(missing)
    p.load_state_dict(sd)
    with pytest.raises(pl.PlannerError, match='exceeds'):
        p.search(np.zeros((5, 3, 4), np.float32), None, 1, 1, 1.0)
    with pytest.raises(pl.PlannerError, match='mz_selfplay_reset'):
        p.selfplay_step(1.0, 1)
    with pytest.raises(pl.PlannerError, match='CartPole'):
        p.selfplay_reset(pl.ENV_CARTPOLE)  # wrong shapes for that env
    r = p.search(obs, None, 1, 1, 1.0)  # the handle is still usable after errors
    assert r['visits'].sum(1).tolist() == [5, 5]


def test_create_rejects_unsupported_configurations():
    from muzero_amd import planner as pl

    net = build_mlp(mlp_case('tiny'))
    with pytest.raises(pl.PlannerError, match='num_simulations'):
        pl.Planner(_cfg(net, num_envs=1, num_simulations=0), 0)
    with pytest.raises(pl.PlannerError, match='discount'):
        pl.Planner(_cfg(net, num_envs=1, num_simulations=3, is_board_game=True, discount=0.9), 0)
    with pytest.raises(pl.PlannerError, match='device_id'):
        pl.Planner(_cfg(net, num_envs=1, num_simulations=3), 99)
    wide = build_mlp(('w', (4,), 257, 16, 1, 1, 16, 1))
    with pytest.raises(pl.PlannerError, match='num_actions'):
        pl.Planner(_cfg(wide, num_envs=1, num_simulations=3), 0)
    atari = build_conv(('a', 'atari', (4, 96, 96), 4, 1, 8, 5, 5, 1))
    c = _cfg(atari, num_envs=1, num_simulations=2)
    c.obs_h = 84
    with pytest.raises(pl.PlannerError, match='96x96'):
        pl.Planner(c, 0)


def test_injected_tie_stream_exhaustion_is_reported():
    """Parity mode with too few recorded tie-break draws: MZ_E_TIES, not a silent default."""
    from muzero_amd import planner as pl

    net = build_mlp(mlp_case('tiny'))
    p = pl.Planner(_cfg(net, num_envs=1, num_simulations=30, max_ties=1, root_dirichlet_alpha=0.0), 0)
    sd = {k: v * 0 for k, v in net.state_dict().items()}  # all-zero net: uniform priors, equal values -> ties at every level
    p.load_state_dict(sd)
    with pytest.raises(pl.PlannerError, match='tie-break stream exhausted'):
        p.search(np.zeros((1, 3, 4), np.float32), None, 1, 1, 1.0, u_tie=np.full((1, 1), 0.3), u_final=0.5)


@pytest.mark.parametrize('shape', ['one_env_one_sim', 'widest_mlp', 'single_legal_action'])
def test_edge_shapes_against_oracle(oracle, shape):
    from muzero_amd import planner as pl
    from test_oracle_nets import _oracle_net

    if shape == 'widest_mlp':
        case, B, S = ('w', (5,), 64, 24, 3, 3, 12, 31), 3, 9   # 64 actions: the LDS kernel's limit
    else:
        case, B, S = mlp_case('tiny'), 1, (1 if shape == 'one_env_one_sim' else 12)
    net = build_mlp(case)
    onet = _oracle_net(oracle, net, 'mlp')
    A = case[2]
    p = pl.Planner(_cfg(net, num_envs=B, num_simulations=S, discount=0.997), 0)
    p.load_state_dict(net.state_dict())
    rs = np.random.RandomState(1)
    obs = rs.uniform(-1, 1, size=(B,) + tuple(case[1])).astype(np.float32)
    mask = np.ones((B, A), bool)
    if shape == 'single_legal_action':
        mask[:] = False
        mask[:, 1] = True
    noise = rs.dirichlet(np.full(A, 0.25), size=B)
    u_tie, u_final = rs.rand(B, 4 * S + 8), rs.rand(B)
    r = p.search(obs, mask, 1, 1, 0.5, False, noise=noise, u_tie=u_tie, u_final=u_final)
    ocfg = oracle.make_config(A, S, 0.997)
    o = oracle.uct_search_batch(ocfg, onet, obs, mask.astype(np.uint8), 1, 1, 0.5, False, noise=noise, u_tie=u_tie, u_final=u_final)
    for k in ('visits', 'pi', 'action', 'root_value'):
        np.testing.assert_array_equal(r[k], o[k])


@pytest.mark.parametrize('shape', ['many_simulations', 'many_actions', 'forced'])
def test_mlp_nets_beyond_the_lds_kernel_use_hbm_trees(oracle, shape, monkeypatch):
    """MLP configurations whose trees do not fit a workgroup's LDS (CartPole net with 150 simulations; 100 actions) fall back
    to HBM-resident trees around batched inference launches instead of being rejected -- same results, bit for bit."""
    from muzero_amd import planner as pl
    from test_oracle_nets import _oracle_net

    if shape == 'many_simulations':
        case, B, S = mlp_case('cartpole'), 21, 150
    elif shape == 'many_actions':
        case, B, S = ('wide', (6,), 100, 48, 5, 5, 24, 41), 19, 20
    else:
        monkeypatch.setenv('MZ_HBM_TREE', '1')
        case, B, S = mlp_case('tictactoe'), 40, 25
    net = build_mlp(case)
    onet = _oracle_net(oracle, net, 'mlp')
    A = case[2]
    board = shape == 'forced'
    kw = dict(num_simulations=S, discount=1.0 if board else 0.997, is_board_game=board, known_bounds=(-1.0, 1.0) if board else None)
    p = pl.Planner(_cfg(net, num_envs=B, **kw), 0)
    p.load_state_dict(net.state_dict())
    rs = np.random.RandomState(3)
    obs = rs.uniform(-1, 1, size=(B,) + tuple(case[1])).astype(np.float32)
    mask = rs.rand(B, A) < 0.8
    mask[np.arange(B), rs.randint(0, A, B)] = True
    cur = rs.randint(1, 3, B).astype(np.int32) if board else np.ones(B, np.int32)
    opp = (3 - cur).astype(np.int32) if board else np.ones(B, np.int32)
    noise = rs.dirichlet(np.full(A, 0.25), size=B)
    u_tie, u_final = rs.rand(B, 4 * S + 8), rs.rand(B)
    r = p.search(obs, mask, cur, opp, 1.0, False, noise=noise, u_tie=u_tie, u_final=u_final)
    ocfg = oracle.make_config(A, S, kw['discount'], board, kw['known_bounds'], 0.25, 0.25)
    o = oracle.uct_search_batch(ocfg, onet, obs, mask.astype(np.uint8), cur, opp, np.ones(B), False, noise=noise, u_tie=u_tie, u_final=u_final)
    for k in ('visits', 'pi', 'action', 'root_value'):
        np.testing.assert_array_equal(r[k], o[k])


def test_attach_replay_refuses_host_memory_through_the_abi():
    """`mz_replay_ring` carries device pointers; a C caller that hands over host arrays gets MZ_E_INVALID naming the field, not a fault inside
    the epilogue kernel (the Python wrapper checks the tensors' device before it ever gets there)."""
    import ctypes as C

    import torch

    from muzero_amd import planner as pl

    net = build_mlp(mlp_case('cartpole'))
    p = pl.Planner(_cfg(net, num_envs=4, num_simulations=5), 0)
    p.load_state_dict(net.state_dict())
    cap, K, A, D = 64, 5, 2, 20
    dev = torch.device('cuda', 0)
    st, ac, pi = torch.zeros(cap, D, device=dev), torch.zeros(cap, K, dtype=torch.int8, device=dev), torch.zeros(cap, K, A, device=dev)
    va, re, pr, cnt = torch.zeros(cap, K, device=dev), torch.zeros(cap, K, device=dev), torch.zeros(cap, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    host_state = torch.zeros(cap, D)  # NOT on the GPU
    r = pl.MzReplayRing(cap, host_state.data_ptr(), ac.data_ptr(), pi.data_ptr(), va.data_ptr(), re.data_ptr(), pr.data_ptr(), cnt.data_ptr(), None, 200, K, 10)
    assert p.lib.mz_selfplay_attach_replay(p.h, C.byref(r)) == -1
    assert b'mz_replay_ring.state is not memory' in p.lib.mz_last_error()
    r = pl.MzReplayRing(cap, st.data_ptr(), ac.data_ptr(), pi.data_ptr(), va.data_ptr(), re.data_ptr(), pr.data_ptr(), cnt.data_ptr(), None, 200, K, 10)
    assert p.lib.mz_selfplay_attach_replay(p.h, C.byref(r)) == 0  # the same ring in HBM is accepted
    p.close()
