"""Evaluator-side pieces: the host board env against the game traces recorded from the reference's
TicTacToeEnv / GomokuEnv (tests/golden/env_cases.npz, incl. the reference tests' win lines), and Elo ratings against
values computed by the reference's rating.py formulas."""
import numpy as np
import pytest

from helpers import load_golden
from muzero_amd.games import BoardGameEnv, GomokuEnv, TicTacToeEnv
from muzero_amd.rating import compute_elo_rating, estimate_win_probability

G = load_golden('env_cases.npz')


def _replay(env, prefix):
    obs = env.reset()
    np.testing.assert_array_equal(obs, G[f'{prefix}_obs'][0])
    assert obs.dtype == np.int8
    np.testing.assert_array_equal(env.actions_mask, G[f'{prefix}_mask'][0].astype(bool))
    assert env.current_player == int(G[f'{prefix}_cur'][0])
    for t, a in enumerate(G[f'{prefix}_actions']):
        obs, r, done, _ = env.step(int(a))
        np.testing.assert_array_equal(obs, G[f'{prefix}_obs'][t + 1])
        assert r == float(G[f'{prefix}_reward'][t]) and done == bool(G[f'{prefix}_done'][t])
        np.testing.assert_array_equal(env.actions_mask, G[f'{prefix}_mask'][t + 1].astype(bool))
        assert env.current_player == int(G[f'{prefix}_cur'][t + 1])
        assert (env.winner or 0) == int(G[f'{prefix}_winner'][t])


@pytest.mark.parametrize('j', range(int(G['ttt_n'])) if 'ttt_n' in G.files else range(40))
def test_tictactoe_traces(j):
    _replay(TicTacToeEnv(), f'ttt_{j}')


@pytest.mark.parametrize('j', range(int(G['gomoku_n'])))
def test_gomoku_traces(j):
    _replay(GomokuEnv(board_size=int(G[f'gomoku_{j}_board'])), f'gomoku_{j}')


def test_env_errors():
    env = TicTacToeEnv()
    with pytest.raises(ValueError):
        env.step(10)
    env.step(4)
    with pytest.raises(ValueError):
        env.step(4)  # taken
    env.step(9)  # white resigns
    assert env.is_game_over and env.winner == 1 and env.loser == 2
    with pytest.raises(RuntimeError):
        env.step(0)
    with pytest.raises(AssertionError):
        BoardGameEnv(black_player_id=1, white_player_id=1)


def test_elo_ratings():
    """rating.py:18-69: expected scores and the K = 32 update, incl. the evaluator's start at -2000 / -2000."""
    assert estimate_win_probability(1500, 1500) == 0.5
    assert abs(estimate_win_probability(1600, 1500) - 1.0 / (1 + 10 ** (-0.25))) < 1e-15
    assert compute_elo_rating(None, 10, 20) == (10, 20)
    a, b = compute_elo_rating(0, -2000, -2000)
    assert (a, b) == (-1984.0, -2016.0)
    a, b = compute_elo_rating(1, 1613, 1609)
    pa = 1.0 / (1 + 10 ** ((1609 - 1613) / 400))
    assert abs(a - (1613 - 32 * pa)) < 1e-12 and abs(b - (1609 + 32 * pa)) < 1e-12
    with pytest.raises(ValueError):
        compute_elo_rating(2, 0, 0)
    with pytest.raises(ValueError):
        compute_elo_rating(0.0, 0, 0)


def test_host_cartpole_matches_oracle_env(oracle):
    """games.CartPoleEnv (evaluator side) against the oracle's CartPole from the same initial states and actions: float64 physics
    with the same operations (identical up to libm's sin/cos, tolerance 1e-6 on float32 rows), stacking, reward, termination."""
    from muzero_amd.games import CartPoleEnv

    rs = np.random.RandomState(8)
    for _ in range(5):
        init = rs.uniform(-0.05, 0.05, size=4)
        env, o = CartPoleEnv(4), oracle.CartPoleEnv(4)
        obs, oobs = env.reset(init), o.reset(init)
        np.testing.assert_allclose(obs, oobs, rtol=1e-6, atol=1e-7)
        assert obs.shape == (4, 5) and obs.dtype == np.float32 and (obs[:, 4] == 0.5).all()
        for t in range(600):
            a = int(rs.randint(0, 2))
            obs, r, done, _ = env.step(a)
            oobs, orw, odone = o.step(a)
            np.testing.assert_allclose(obs, oobs, rtol=1e-6, atol=1e-7)
            assert r == 1.0 == orw and done == odone
            if done:
                break
        assert done and env.steps <= 500
        with pytest.raises(RuntimeError):
            env.step(0)
    assert env.actions_mask.all() and env.current_player == env.opponent_player == 1
