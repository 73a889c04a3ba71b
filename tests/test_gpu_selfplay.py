"""Device-resident self-play (mz_selfplay_*: search + sample + env.step + record + auto-reset in HBM) on an MI355X:
the recorded trajectories replayed through the oracle environments, and the run_self_play counterpart end to end."""
import queue
import types

import numpy as np
import pytest

from helpers import build_mlp, mlp_case

pytestmark = pytest.mark.gpu


def _planner(net, num_envs, seed=3, **search):
    from muzero_amd import planner as pl

    p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=num_envs, seed=seed, **search), 0)
    p.load_state_dict(net.state_dict())
    return p


def test_tictactoe_device_env_matches_oracle_env(oracle):
    """Integer board logic: every recorded (obs, player, reward, done) equals the oracle BoardEnv replaying the recorded
    actions, across auto-resets; actions are legal; policies are distributions over legal moves."""
    from muzero_amd import planner as pl

    net = build_mlp(mlp_case('tictactoe'))
    B, M = 48, 40
    p = _planner(net, B, num_simulations=25, discount=1.0, is_board_game=True, known_bounds=(-1.0, 1.0))
    p.selfplay_reset(pl.ENV_TICTACTOE)
    p.selfplay_step(-1.0, M)
    rec = p.selfplay_read(M)
    cnt = p.selfplay_counters()
    assert cnt['env_steps'] == B * M and cnt['simulations'] == B * M * 25
    assert cnt['episodes'] == int(rec['done'].sum()) and cnt['episode_steps'] > 0
    for b in range(B):
        env = oracle.BoardEnv(3, 4, 3)
        obs = env.reset()
        for m in range(M):
            np.testing.assert_array_equal(rec['obs'][m, b].reshape(9, 3, 3), obs.astype(np.float32))
            assert rec['player'][m, b] == env.current_player
            a = int(rec['action'][m, b])
            assert env.actions_mask[a], 'sampled action must be legal'
            pi = rec['pi'][m, b]
            assert abs(pi.sum() - 1.0) < 1e-12 and (pi[~env.actions_mask] == 0).all()
            obs, r, done = env.step(a)
            assert r == rec['reward'][m, b] and done == bool(rec['done'][m, b])
            if done:
                obs = env.reset()
    assert rec['done'].sum() > B  # several finished games per env slot on average


def test_cartpole_device_env_matches_oracle_env(oracle):
    """float64 physics + float32 stacked observations vs the oracle CartPole (one ulp of the device sin/cos is allowed for:
    1e-6 relative on float32 observations), ACROSS auto-resets: episode 0 starts from injected states, every later episode
    from the device's Philox reset stream U(-0.05, 0.05)^4 keyed by (seed, env, episode), which the test regenerates."""
    from helpers import philox_uniforms
    from muzero_amd import planner as pl

    net = build_mlp(mlp_case('cartpole'))
    B, M, seed = 48, 120, 3
    rs = np.random.RandomState(4)
    init = rs.uniform(-0.05, 0.05, size=(B, 4))
    p = _planner(net, B, seed=seed, num_simulations=50, discount=0.997)
    p.selfplay_reset(pl.ENV_CARTPOLE, init)
    rec = {}
    for lo in range(0, M, 40):  # the record ring keeps 64 moves
        p.selfplay_step(1.0, 40)
        part = p.selfplay_read(40)
        for k, v in part.items():
            rec[k] = v if k not in rec else np.concatenate([rec[k], v])
    assert (rec['reward'] == 1.0).all() and (rec['player'] == 1).all()
    np.testing.assert_array_equal(rec['pi'], np.round(rec['pi'] * 50) / 50)  # T = 1: visit counts / 50
    episodes = 0
    for b in range(B):
        env = oracle.CartPoleEnv(4)
        obs = env.reset(init[b])
        ep = 0
        for m in range(M):
            np.testing.assert_allclose(rec['obs'][m, b].reshape(4, 5), obs, rtol=1e-6, atol=1e-7)
            obs, r, done = env.step(int(rec['action'][m, b]))
            assert done == bool(rec['done'][m, b]), (b, m)
            if done:
                ep += 1
                obs = env.reset(-0.05 + 0.1 * philox_uniforms(seed, b, ep, 0x40000000, 4))  # mz_env.h cartpole_fresh
        episodes += ep
    assert episodes > 2 * B  # random-weight policies drop the pole within a few dozen steps


def test_cartpole_time_limit_truncation(oracle):
    """TimeLimit 500 (gym registration of CartPole-v1; gym_env.py:452): with the step counters moved to 497 the third step
    from there ends every surviving episode with done = 1 and the env resets to its Philox state."""
    import ctypes as C
    from helpers import philox_uniforms
    from muzero_amd import planner as pl

    net = build_mlp(mlp_case('cartpole'))
    B, seed = 32, 9
    init = np.zeros((B, 4))  # upright and centred: no env fails within the few steps played
    p = _planner(net, B, seed=seed, num_simulations=50, discount=0.997)
    p.selfplay_reset(pl.ENV_CARTPOLE, init)
    p.selfplay_step(1.0, 2)
    p.lib.mz_debug_set_env_steps.argtypes = [C.c_void_p, C.c_void_p]
    steps = np.full(B, 497, np.int32)
    assert p.lib.mz_debug_set_env_steps(p.h, steps.ctypes.data_as(C.c_void_p)) == 0
    p.selfplay_step(1.0, 4)
    rec = p.selfplay_read(6)
    np.testing.assert_array_equal(rec['done'][:, :].astype(int), np.array([0, 0, 0, 0, 1, 0])[:, None] * np.ones((1, B), int))
    for b in range(B):
        fresh = (-0.05 + 0.1 * philox_uniforms(seed, b, 1, 0x40000000, 4)).astype(np.float32)
        o = rec['obs'][5, b].reshape(4, 5)
        np.testing.assert_array_equal(o[:, :4], np.tile(fresh, (4, 1)))
        np.testing.assert_array_equal(o[:, 4], np.full(4, np.float32(0.5)))


def test_run_self_play_counterpart_emits_reference_shaped_items():
    """pipeline.run_self_play: (Transition, priority) stream with the reference's shapes/dtypes and MC-return values."""
    import torch
    from muzero_amd import pipeline
    from muzero_amd.config import make_tictactoe_config

    net = build_mlp(mlp_case('tictactoe'))
    cfg = make_tictactoe_config(use_tensorboard=False)
    cfg.num_envs = 32
    q = queue.SimpleQueue()
    stop = types.SimpleNamespace(is_set=lambda: False)
    steps = pipeline.run_self_play(cfg, 0, net, torch.device('cuda', 0), 'TicTacToe', q, types.SimpleNamespace(value=0), stop, max_moves=24)
    assert steps == 24 * 32
    items = []
    while not q.empty():
        items.append(q.get())
    assert len(items) > 32
    for tr, prio in items:
        assert tr.state.shape == (9, 3, 3) and tr.action.shape == (5,) and tr.action.dtype == np.int8
        assert tr.pi_prob.shape == (5, 10) and tr.pi_prob.dtype == np.float32
        assert tr.value.dtype == np.float32 and tr.reward.dtype == np.float32
        assert set(np.unique(tr.value)).issubset({-1.0, 0.0, 1.0})  # Monte-Carlo returns of a board game
        assert np.isfinite(prio) and prio >= 0


def test_gomoku_device_env_matches_oracle_env(oracle):
    """Gomoku (9x9 board, five in a row, A = 82) on a conv planner: the recorded trajectories replayed through the oracle
    BoardEnv -- bit-exact integer board logic across auto-resets."""
    from helpers import build_conv, conv_case
    from muzero_amd import planner as pl

    net = build_conv(conv_case('board9'))
    B, M = 24, 60
    p = _planner(net, B, num_simulations=12, discount=1.0, is_board_game=True, known_bounds=(-1.0, 1.0), root_dirichlet_alpha=0.03)
    p.selfplay_reset(pl.ENV_GOMOKU)
    done_total = 0
    envs = [oracle.BoardEnv(9, 4, 5) for _ in range(B)]
    obs = [e.reset() for e in envs]
    steps = [0] * B
    for _ in range(M // 12):
        p.selfplay_step(-1.0, 12)
        rec = p.selfplay_read(12)
        for m in range(12):
            for b in range(B):
                env = envs[b]
                np.testing.assert_array_equal(rec['obs'][m, b].reshape(9, 9, 9), obs[b].astype(np.float32))
                assert rec['player'][m, b] == env.current_player
                a = int(rec['action'][m, b])
                assert env.actions_mask[a], 'sampled action must be legal'
                pi = rec['pi'][m, b]
                assert abs(pi.sum() - 1.0) < 1e-12 and (pi[~env.actions_mask] == 0).all()
                obs[b], r, done = env.step(a)
                steps[b] += 1
                assert r == rec['reward'][m, b] and done == bool(rec['done'][m, b])
                if done:
                    done_total += 1
                    obs[b] = env.reset()
                    steps[b] = 0
    cnt = p.selfplay_counters()
    assert cnt['env_steps'] == B * M and cnt['episodes'] == done_total


def test_synthetic_frame_env_runs_atari_net():
    """The Atari stand-in env: observations are fresh U[0,1) frames each step, reward 0, search output is a distribution."""
    from helpers import build_conv, conv_case
    from muzero_amd import planner as pl

    net = build_conv(conv_case('atari_s'))
    B, M = 5, 3
    p = _planner(net, B, num_simulations=4, discount=0.997)
    p.selfplay_reset(pl.ENV_SYNTHETIC)
    p.selfplay_step(1.0, M)
    rec = p.selfplay_read(M)
    assert (rec['reward'] == 0).all() and (rec['done'] == 0).all() and (rec['player'] == 1).all()
    assert rec['obs'].min() >= 0.0 and rec['obs'].max() < 1.0 and 0.45 < rec['obs'].mean() < 0.55
    assert not np.array_equal(rec['obs'][0], rec['obs'][1])
    np.testing.assert_allclose(rec['pi'].sum(-1), 1.0, atol=1e-12)
    np.testing.assert_array_equal(rec['pi'], np.round(rec['pi'] * 4) / 4)


def test_synthetic_env_with_mlp_net_redraws_the_observation_every_move():
    """with the fused MLP move as the default the synthetic env's frame kernel was skipped -- every move searched the
    same stale frame.  The LunarLander-shaped bench leg runs exactly this combination."""
    from muzero_amd import planner as pl

    net = build_mlp(mlp_case('lunar'))
    B, M = 40, 4
    p = _planner(net, B, num_simulations=6, discount=0.997)
    p.selfplay_reset(pl.ENV_SYNTHETIC)
    p.selfplay_step(1.0, M)
    rec = p.selfplay_read(M)
    assert (rec['reward'] == 0).all() and (rec['done'] == 0).all()
    for m in range(1, M):
        assert not np.array_equal(rec['obs'][m], rec['obs'][m - 1])
    assert rec['obs'].min() >= 0.0 and rec['obs'].max() < 1.0
    np.testing.assert_allclose(rec['pi'].sum(-1), 1.0, atol=1e-12)
    assert set(np.unique(rec['action'])).issubset({0, 1, 2, 3})


def test_run_self_play_gomoku_conv_net_emits_mc_return_items():
    """pipeline.run_self_play on the device Gomoku env with a conv (board) network: reference-shaped items, Monte-Carlo
    returns in {-1, 0, 1}, observations are the 9-plane board stacks."""
    import torch
    from helpers import build_conv, conv_case
    from muzero_amd import pipeline
    from muzero_amd.config import make_gomoku_config

    net = build_conv(conv_case('board9'))
    cfg = make_gomoku_config(use_tensorboard=False)
    cfg.num_envs, cfg.num_simulations = 16, 8
    q = queue.SimpleQueue()
    stop = types.SimpleNamespace(is_set=lambda: False)
    steps = pipeline.run_self_play(cfg, 0, net, torch.device('cuda', 0), 'Gomoku', q, types.SimpleNamespace(value=0), stop, max_moves=96,
                                   moves_per_drain=16)
    assert steps == 96 * 16
    items = []
    while not q.empty():
        items.append(q.get())
    assert len(items) > 16  # at least one finished game per env slot on average (81 points, random-ish play)
    for tr, prio in items:
        assert tr.state.shape == (9, 9, 9) and set(np.unique(tr.state)).issubset({0.0, 1.0})
        assert tr.action.shape == (5,) and tr.pi_prob.shape == (5, 82) and tr.value.shape == (5,)
        assert set(np.unique(tr.value)).issubset({-1.0, 0.0, 1.0}) and np.isfinite(prio)


@pytest.mark.parametrize('game', ['cartpole', 'tictactoe'])
def test_selfplay_search_outputs_equal_oracle_search(oracle, game):
    """the SEARCH a device self-play move runs (pipeline.py:95-113) -- for TicTacToe the FUSE = true
    instantiation of k_search_fast, which is the C3 bench kernel, for CartPole the two-action one behind k_env_pre / k_env_step
    -- compared with the oracle, not just checked for being a distribution: every move's recorded observation, legal mask,
    player to move and temperature, plus the Philox draws that move consumed (mz_debug_capture_rng), go through
    oracle.uct_search_batch; policy, root value and sampled action must be EQUAL.  >= 256 envs x >= 40 moves, across auto-resets."""
    import os

    B, M = int(os.environ.get('MZ_SELFPLAY_B', '256')), int(os.environ.get('MZ_SELFPLAY_M', '40'))  # (soak runs: 4096 x 100)
    selfplay_search_vs_oracle(oracle, game, mlp_case(game), 25 if game == 'tictactoe' else 50, B, M)


def selfplay_search_vs_oracle(oracle, game, case, S, B, M, seed=77, expect_resets=True, reload_case=None):
    """The comparison of test_selfplay_search_outputs_equal_oracle_search for any MLP net that fits `game`'s observations and actions."""
    import ctypes as C

    from test_oracle_nets import _oracle_net
    from muzero_amd import planner as pl

    board = game in ('tictactoe', 'gomoku')
    conv = game == 'gomoku'  # `case`: a conv case tuple (helpers.CONV_CASES layout), observations (2 * stack + 1, N, N)
    if conv:
        from helpers import build_conv

        net = build_conv(case)
        A, obs_shape = case[3], tuple(case[2])
        N, stack = obs_shape[1], (obs_shape[0] - 1) // 2
    else:
        net = build_mlp(case)
        A, obs_shape = case[2], tuple(case[1])
    kw = dict(num_simulations=S, discount=1.0 if board else 0.997, is_board_game=board, known_bounds=(-1.0, 1.0) if board else None,
              root_dirichlet_alpha=0.25, root_exploration_eps=0.25)
    p = _planner(net, B, seed=seed, **kw)
    p.lib.mz_debug_capture_rng.argtypes = [C.c_void_p, C.c_int32]
    p.lib.mz_debug_read_rng.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert p.lib.mz_debug_capture_rng(p.h, 1) == 0
    p.selfplay_reset(pl.ENV_GOMOKU if conv else (pl.ENV_TICTACTOE if board else pl.ENV_CARTPOLE))
    onet = _oracle_net(oracle, net, 'conv' if conv else 'mlp')
    cfg = oracle.make_config(A, S, kw['discount'], board, kw['known_bounds'], 0.25, 0.25)
    t_switch = 30 if conv else 6  # env steps at temperature 1 (config.py:236-249)
    envs = [oracle.BoardEnv(N, stack, 5) if conv else oracle.BoardEnv(3, 4, 3) for _ in range(B)] if board else None
    if board:
        for e in envs:
            e.reset()
    steps = np.zeros(B, np.int64)  # env steps of the running episode (the board games' temperature schedule, config.py:236-241)
    finished = 0
    for m in range(M):
        if reload_case is not None and m == M // 2:
            # new weights in the middle of self-play (what run_self_play does when the learner publishes a checkpoint, pipeline.py:261-267)
            net = build_conv(reload_case) if conv else build_mlp(reload_case)
            p.load_state_dict(net.state_dict())
            onet = _oracle_net(oracle, net, 'conv' if conv else 'mlp')
        p.selfplay_step(-1.0 if board else 1.0, 1)
        noise = np.empty((B, A), np.float64)
        utie = np.empty((B, p.max_ties), np.float64)
        ufin = np.empty(B, np.float64)
        assert p.lib.mz_debug_read_rng(p.h, noise.ctypes.data_as(C.c_void_p), utie.ctypes.data_as(C.c_void_p), ufin.ctypes.data_as(C.c_void_p)) == 0
        rec = p.selfplay_read(1)
        obs = rec['obs'][0].reshape((B,) + obs_shape)
        if board:
            mask = np.stack([e.actions_mask for e in envs]).astype(np.uint8)
            cur = np.array([e.current_player for e in envs], np.int32)
            np.testing.assert_array_equal(rec['player'][0], cur)
            opp = 3 - cur
            T = np.where(steps < t_switch, 1.0, 0.1)
        else:
            mask, cur, opp, T = np.ones((B, A), np.uint8), 1, 1, 1.0
        o = oracle.uct_search_batch(cfg, onet, obs, mask, cur, opp, T, False, noise=noise, u_tie=utie, u_final=ufin)
        # the documented deviation of production self-play: when every root visit fell on illegal actions (possible with
        # very few simulations) the reference's policy is 0 / 0 and np.random.choice raises; the device plays uniformly over the legal moves
        bad = np.isnan(o['pi']).any(axis=1)
        ok = ~bad
        np.testing.assert_array_equal(rec['pi'][0][ok], o['pi'][ok], err_msg=f'move {m}: policy')
        np.testing.assert_array_equal(rec['root_value'][0][ok], o['root_value'][ok], err_msg=f'move {m}: root value')
        np.testing.assert_array_equal(rec['action'][0][ok], o['action'][ok], err_msg=f'move {m}: action')
        for b in np.flatnonzero(bad):
            legal = np.asarray(mask[b], bool)
            np.testing.assert_allclose(rec['pi'][0][b], legal / legal.sum(), rtol=0, atol=1e-15)
            assert legal[int(rec['action'][0][b])]
        done = rec['done'][0].astype(bool)
        finished += int(done.sum())
        steps = np.where(done, 0, steps + 1)
        if board:
            for b in range(B):
                _, _, d = envs[b].step(int(rec['action'][0, b]))
                assert d == bool(done[b])
                if d:
                    envs[b].reset()
    if expect_resets:
        assert finished > (B if board else B // 2)  # the comparison ran across auto-resets
    p.close()
