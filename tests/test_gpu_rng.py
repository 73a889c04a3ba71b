"""The PRODUCTION randomness of the planner -- the on-device Philox streams the self-play bench runs on (root Dirichlet
noise mcts.py:245, tie-break draws mcts.py:124, final action sample mcts.py:404) -- which the parity tests bypass by
injecting recorded draws.  A test hook (mz_debug_capture_rng / mz_debug_read_rng, not part of the ABI) returns the draws a
Philox-mode search consumed, in the layout of the injected inputs:
  * their distributions are checked (Dirichlet moments for alpha in {0.03, 0.25} x A in {2, 10, 226}; uniforms);
  * the same search replayed in parity mode with the captured draws gives identical results -- on the GPU and on the
    oracle, which closes the loop between the code path the bench runs and the reference-pinned checker."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import build_mlp, mlp_case, seeded_state_dict

pytestmark = pytest.mark.gpu


def _capture(p, on=True):
    p.lib.mz_debug_capture_rng.argtypes = [C.c_void_p, C.c_int32]
    p.lib.mz_debug_read_rng.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert p.lib.mz_debug_capture_rng(p.h, 1 if on else 0) == 0


def _read(p):
    noise = np.empty((p.B, p.A), np.float64)
    utie = np.empty((p.B, p.max_ties), np.float64)
    ufin = np.empty(p.B, np.float64)
    assert p.lib.mz_debug_read_rng(p.h, noise.ctypes.data_as(C.c_void_p), utie.ctypes.data_as(C.c_void_p), ufin.ctypes.data_as(C.c_void_p)) == 0
    return noise, utie, ufin


def _beta_moments(a, b):
    """mean, variance and the standard error factor of the sample variance (sqrt(m4 - var^2)) of Beta(a, b)."""
    raw = [1.0]
    for k in range(4):
        raw.append(raw[-1] * (a + k) / (a + b + k))
    m = raw[1]
    var = raw[2] - m * m
    m4 = raw[4] - 4 * m * raw[3] + 6 * m * m * raw[2] - 3 * m ** 4
    return m, var, np.sqrt(max(m4 - var * var, 0.0))


@pytest.mark.parametrize('A,alpha', [(2, 0.25), (2, 0.03), (10, 0.25), (10, 0.03), (226, 0.25), (226, 0.03)])
def test_production_dirichlet_noise_moments(A, alpha):
    """Each component of Dirichlet(alpha 1_A) is Beta(alpha, (A - 1) alpha): mean 1/A, variance (A - 1) / (A^2 (A alpha + 1)).
    >= 1e5 root noise vectors per case from the Philox gamma sampler (mz_device.h gamma_sample, mz_search.h root_noise_lanes)."""
    from muzero_amd import network, planner as pl

    net = network.MuZeroMLPNet((3, 4), A, 32, 7, 5, 16)
    net.load_state_dict(seeded_state_dict(net, 77))
    net.eval()
    B = 4096
    moves = 25
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=B, seed=1234 + A, num_simulations=2, discount=0.997,
                                     root_dirichlet_alpha=alpha, root_exploration_eps=0.25), 0)
    p.load_state_dict(net.state_dict())
    _capture(p)
    rs = np.random.RandomState(0)
    obs = rs.uniform(-1, 1, size=(B, 3, 4)).astype(np.float32)
    draws = []
    for _ in range(moves):
        p.search(obs, np.ones((B, A), bool), 1, 1, 1.0, False)
        draws.append(_read(p)[0].copy())
    x = np.concatenate(draws)  # [N, A]
    N = x.shape[0]
    assert N >= 100000
    np.testing.assert_allclose(x.sum(axis=1), 1.0, rtol=0, atol=1e-12)
    assert (x >= 0).all()
    assert not np.array_equal(draws[0], draws[1])  # a fresh stream per move
    mean, var, se4 = _beta_moments(alpha, (A - 1) * alpha)
    assert abs(mean - 1.0 / A) < 1e-15 and abs(var - (A - 1) / (A * A * (A * alpha + 1))) < 1e-15
    # per-component means: 6 sigma of the sample mean; the variance pooled over the components, 6 sigma of one component's
    # sample variance (the components are exchangeable, so pooling only tightens it)
    assert np.abs(x.mean(axis=0) - mean).max() < 6.0 * np.sqrt(var / N)
    assert abs(x.var(axis=0).mean() - var) < 6.0 * se4 / np.sqrt(N)
    # the mass near the corners is what alpha < 1 is about: P(component > 0.5) of Beta(alpha, (A-1) alpha)
    from scipy import stats

    p_hi = stats.beta.sf(0.5, alpha, (A - 1) * alpha)
    got = (x > 0.5).mean()
    assert abs(got - p_hi) < 6.0 * np.sqrt(p_hi * (1 - p_hi) / (N * A)) + 1e-6
    p.close()


def test_production_tie_and_final_draws_are_uniform():
    """The first simulation of every search ties all A actions (U == 0 at an unvisited root, mcts.py:193-195): its draw picks
    uniformly.  10 actions, 4096 envs x 25 moves."""
    from muzero_amd import network, planner as pl
    from scipy import stats

    A, B, moves = 10, 4096, 25
    net = network.MuZeroMLPNet((3, 4), A, 32, 7, 5, 16)
    net.load_state_dict(seeded_state_dict(net, 78))
    net.eval()
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=B, seed=99, num_simulations=3, discount=0.997,
                                     root_dirichlet_alpha=0.25, root_exploration_eps=0.25), 0)
    p.load_state_dict(net.state_dict())
    _capture(p)
    obs = np.random.RandomState(1).uniform(-1, 1, size=(B, 3, 4)).astype(np.float32)
    first, final, acts, pis = [], [], [], []
    for _ in range(moves):
        r = p.search(obs, np.ones((B, A), bool), 1, 1, 1.0, False)
        _, utie, ufin = _read(p)
        first.append(utie[:, 0].copy())
        final.append(ufin.copy())
        acts.append(r['action'].copy())
        pis.append(r['pi'].copy())
    u = np.concatenate(first)
    assert ((u >= 0) & (u < 1)).all()
    assert stats.kstest(u, 'uniform').pvalue > 1e-4
    picks = np.floor(u * A).astype(int)
    assert stats.chisquare(np.bincount(picks, minlength=A)).pvalue > 1e-4
    uf = np.concatenate(final)
    assert stats.kstest(uf, 'uniform').pvalue > 1e-4
    # and the sampled action is the inverse-CDF of pi at that uniform (np.random.choice(p=pi), mcts.py:404)
    pi = np.concatenate(pis)
    cdf = np.cumsum(pi, axis=1)
    cdf /= cdf[:, -1:]
    expect = np.minimum((cdf <= uf[:, None]).sum(axis=1), A - 1)
    np.testing.assert_array_equal(np.concatenate(acts), expect)
    p.close()


@pytest.mark.parametrize('game', ['cartpole', 'tictactoe'])
def test_production_search_replays_bit_exact_in_parity_mode(oracle, game):
    """Philox-mode search == injected-mode search == oracle search when the latter two are fed the draws the first one
    consumed: the code path the bench measures is the same arithmetic as the reference-pinned one."""
    from test_oracle_nets import _oracle_net
    from muzero_amd import planner as pl

    board = game == 'tictactoe'
    case = mlp_case(game)
    net = build_mlp(case)
    A, S, B = case[2], (25 if board else 50), 300
    kw = dict(num_simulations=S, discount=1.0 if board else 0.997, is_board_game=board, known_bounds=(-1.0, 1.0) if board else None,
              root_dirichlet_alpha=0.25, root_exploration_eps=0.25)
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=B, seed=4321, **kw), 0)
    p.load_state_dict(net.state_dict())
    _capture(p)
    rs = np.random.RandomState(2)
    obs = rs.uniform(-1, 1, size=(B,) + tuple(case[1])).astype(np.float32)
    mask = np.ones((B, A), bool)
    if board:
        mask[:, :] = rs.rand(B, A) < 0.7
        mask[:, -1] = True
    cur, opp = 1, (2 if board else 1)
    prod = p.search(obs, mask, cur, opp, 1.0, False)
    noise, utie, ufin = _read(p)
    assert (utie[:, 0] != 0.5).all()  # every search drew at least its first tie-break
    inj = p.search(obs, mask, cur, opp, 1.0, False, noise=noise, u_tie=utie, u_final=ufin)
    for k in ('action', 'pi', 'root_value', 'visits'):
        np.testing.assert_array_equal(prod[k], inj[k], err_msg=k)
    onet = _oracle_net(oracle, net, 'mlp')
    cfg = oracle.make_config(A, S, kw['discount'], board, kw['known_bounds'], 0.25, 0.25)
    o = oracle.uct_search_batch(cfg, onet, obs, mask.astype(np.uint8), cur, opp, 1.0, False, noise=noise, u_tie=utie, u_final=ufin)
    for k in ('action', 'pi', 'root_value', 'visits'):
        np.testing.assert_array_equal(prod[k], o[k], err_msg=k)
    p.close()
