"""G5: board-game environment semantics (games/env.py, games/tictactoe.py, games/gomoku.py) -- oracle vs traces
recorded from the reference envs, including the reference tests' 8 win lines x 2 colours
(tests/games/tictactoe_test.py:25-34), resign (tests/games/boardgame_test.py:42-55) and Gomoku five-in-row."""
import numpy as np
import pytest

from helpers import load_golden

G = load_golden('env_cases.npz')


def _replay(oracle, prefix, board_size, stack, num_to_win):
    env = oracle.BoardEnv(board_size, stack, num_to_win)
    obs = env.reset()
    np.testing.assert_array_equal(obs, G[f'{prefix}_obs'][0])
    np.testing.assert_array_equal(env.actions_mask, G[f'{prefix}_mask'][0].astype(bool))
    assert env.current_player == int(G[f'{prefix}_cur'][0])
    for t, a in enumerate(G[f'{prefix}_actions']):
        obs, r, done = env.step(int(a))
        np.testing.assert_array_equal(obs, G[f'{prefix}_obs'][t + 1])
        assert r == float(G[f'{prefix}_reward'][t])
        assert done == bool(G[f'{prefix}_done'][t])
        np.testing.assert_array_equal(env.actions_mask, G[f'{prefix}_mask'][t + 1].astype(bool))
        assert env.current_player == int(G[f'{prefix}_cur'][t + 1])
        assert env.winner == int(G[f'{prefix}_winner'][t])
    return env


@pytest.mark.parametrize('j', range(int(G['ttt_n'])))
def test_tictactoe_trace(oracle, j):
    _replay(oracle, f'ttt_{j}', 3, 4, 3)


@pytest.mark.parametrize('j', range(int(G['gomoku_n'])))
def test_gomoku_trace(oracle, j):
    _replay(oracle, f'gomoku_{j}', int(G[f'gomoku_{j}_board']), 4, 5)


def test_invalid_moves_rejected(oracle):
    # games/env.py:119-124: out of range, already taken, game over
    env = oracle.BoardEnv(3, 4, 3)
    env.step(4)
    with pytest.raises(ValueError):
        env.step(4)
    with pytest.raises(ValueError):
        env.step(10)
    env.step(9)  # resign
    with pytest.raises(ValueError):
        env.step(0)


def test_cartpole_equations(oracle):
    """gym 0.23.1 CartPole-v1 (un-vendored; parity unpinned upstream): independent float64 restatement of the
    published Euler update, plus stacker layout (gym_env.py:306-353)."""
    env = oracle.CartPoleEnv(stack=4)
    s = np.array([0.01, -0.02, 0.03, 0.04])
    obs = env.reset(s)
    assert obs.shape == (4, 5)
    np.testing.assert_array_equal(obs[:, :4], np.tile(s.astype(np.float32), (4, 1)))
    np.testing.assert_array_equal(obs[:, 4], np.full(4, np.float32(0.5)))
    x, xd, th, thd = s
    hist = [obs[0].copy()]
    for t, a in enumerate([1, 0, 1, 1, 0, 0, 1]):
        force = 10.0 if a == 1 else -10.0
        temp = (force + 0.05 * thd * thd * np.sin(th)) / 1.1
        thacc = (9.8 * np.sin(th) - np.cos(th) * temp) / (0.5 * (4.0 / 3.0 - 0.1 * np.cos(th) ** 2 / 1.1))
        xacc = temp - 0.05 * thacc * np.cos(th) / 1.1
        x, xd, th, thd = x + 0.02 * xd, xd + 0.02 * xacc, th + 0.02 * thd, thd + 0.02 * thacc
        obs, r, done = env.step(a)
        assert r == 1.0 and not done
        np.testing.assert_allclose(obs[0, :4], np.array([x, xd, th, thd], np.float32), rtol=1e-6)
        assert obs[0, 4] == np.float32((a + 1) / 2)
        np.testing.assert_array_equal(obs[1], hist[-1])  # newest first
        hist.append(obs[0].copy())
    # termination: pole angle beyond 12 degrees, and TimeLimit 500
    env.reset(np.array([0.0, 0.0, 0.2, 0.0]))
    _, _, done = env.step(0)
    while not done:
        _, _, done = env.step(0)
    assert abs(env.e.s[2]) > 12 * 2 * np.pi / 360 or abs(env.e.s[0]) > 2.4
