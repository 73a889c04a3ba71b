"""G4: helper functions of the path -- oracle vs values recorded from the reference, including the reference's
own known-answer tests (tests/pipeline_test.py:24-53, tests/util_test.py:25-48)."""
import numpy as np
import pytest

from helpers import load_golden

G = load_golden('pipe_cases.npz')


def test_n_step_target_reference_kats(oracle):
    # tests/pipeline_test.py:24-33 -- expected [4.97, 3.982, 2.991, 1.997, 1.0] to 3 decimals
    out = oracle.n_step_target(np.ones(5), np.zeros(5), 5, 0.997)
    np.testing.assert_almost_equal(out, np.array([4.97, 3.982, 2.991, 1.997, 1.0]), decimal=3)
    np.testing.assert_array_equal(out, G['nstep_kat1_out'])
    # tests/pipeline_test.py:35-53
    rv = np.array([0.1 * (i + 1) for i in range(10)])
    out = oracle.n_step_target(np.ones(10), rv, 5, 0.997)
    exp = [4.97 + 0.997**5 * rv[i + 5] for i in range(5)] + [4.97, 3.982, 2.991, 1.997, 1.0]
    np.testing.assert_almost_equal(out, np.array(exp), decimal=3)
    np.testing.assert_array_equal(out, G['nstep_kat2_out'])


@pytest.mark.parametrize('j', range(int(G['nstep_n'])))
def test_n_step_target(oracle, j):
    out = oracle.n_step_target(G[f'nstep_{j}_rewards'], G[f'nstep_{j}_roots'], int(G[f'nstep_{j}_td']), float(G[f'nstep_{j}_discount']))
    np.testing.assert_array_equal(out, G[f'nstep_{j}_out'])


@pytest.mark.parametrize('j', range(int(G['mc_n'])))
def test_mc_return_target(oracle, j):
    out = oracle.mc_return_target(G[f'mc_{j}_rewards'], G[f'mc_{j}_players'])
    np.testing.assert_array_equal(out, G[f'mc_{j}_out'])


@pytest.mark.parametrize('j', range(int(G['unroll_n'])))
def test_make_unroll_sequence(oracle, j):
    st, ac, rw, vl, pi, pr = oracle.make_unroll_sequence(
        list(G[f'unroll_{j}_obs']), G[f'unroll_{j}_actions'], G[f'unroll_{j}_rewards'], list(G[f'unroll_{j}_pis']),
        G[f'unroll_{j}_values'], G[f'unroll_{j}_prios'], 5,
    )
    np.testing.assert_array_equal(st, G[f'unroll_{j}_out_state'])
    np.testing.assert_array_equal(ac, G[f'unroll_{j}_out_action'])
    assert ac.dtype == np.int8
    np.testing.assert_array_equal(rw, G[f'unroll_{j}_out_reward'])
    np.testing.assert_array_equal(vl, G[f'unroll_{j}_out_value'])
    np.testing.assert_array_equal(pi, G[f'unroll_{j}_out_pi'])
    np.testing.assert_array_equal(pr, G[f'unroll_{j}_out_prio'])


@pytest.mark.parametrize('j', range(int(G['policy_n'])))
def test_generate_play_policy(oracle, j):
    T = float(G[f'policy_{j}_T'])
    out = oracle.generate_play_policy(G[f'policy_{j}_visits'], T)
    exponent = max(1.0, min(5.0, 1.0 / T)) if T > 0 else 1.0
    if exponent == int(exponent):
        # every schedule the reference ships (config.py:236-267) gives exponent 1, 2, 4 or 5: exact powers
        np.testing.assert_array_equal(out, G[f'policy_{j}_out'])
    else:
        # numpy's vectorised float64 pow and libm pow may differ in the last bit for non-integer exponents
        np.testing.assert_allclose(out, G[f'policy_{j}_out'], rtol=1e-15, atol=0)


@pytest.mark.parametrize('j', range(int(G['noise_n'])))
def test_root_prior_noise_and_mask(oracle, j):
    p64, _ = oracle.prepare_root_prior(G[f'noise_{j}_p'], G[f'noise_{j}_noise'], 0.25, G[f'noise_{j}_mask'], False)
    np.testing.assert_array_equal(p64, G[f'noise_{j}_masked'])
    p64u, _ = oracle.prepare_root_prior(G[f'noise_{j}_p'], G[f'noise_{j}_noise'], 0.25, None, False)
    np.testing.assert_array_equal(p64u, G[f'noise_{j}_noised'])
    _, p32 = oracle.prepare_root_prior(G[f'noise_{j}_p'], None, 0.25, G[f'noise_{j}_mask'], True)
    np.testing.assert_array_equal(p32, G[f'noise_{j}_masked32'])


def test_signed_parabolic(oracle):
    out = oracle.signed_parabolic(G['xform_x'])
    ref = G['xform_parabolic']
    # the expression cancels catastrophically in float32 (sqrt(.)/2/eps - 500): one ulp of the sqrt is 3e-5 absolute
    np.testing.assert_allclose(out, ref, rtol=2e-4, atol=1.5e-4)


@pytest.mark.parametrize('j', range(4))
def test_logits_to_transformed_expected_value(oracle, j):
    out = oracle.logits_to_value(G[f'logits_{j}_in'])
    np.testing.assert_allclose(out, G[f'logits_{j}_out'].reshape(-1), rtol=2e-4, atol=1.5e-4)


def test_normalize_hidden_state(oracle):
    for row_in, row_out in zip(G['norm_mlp_in'], G['norm_mlp_out']):
        np.testing.assert_array_equal(oracle.normalize_hidden(row_in), row_out)
    for x, y in zip(G['norm_conv_in'], G['norm_conv_out']):
        np.testing.assert_array_equal(oracle.normalize_hidden(x), y)


def test_expf_accuracy(oracle):
    xs = np.concatenate([np.linspace(-100, 0, 4001), np.linspace(0, 80, 801)]).astype(np.float32)
    got = np.array([oracle.lib().mzo_expf(float(x)) for x in xs], np.float64)
    ref = np.exp(xs.astype(np.float64))
    ok = ref > 1e-37
    assert np.max(np.abs(got[ok] / ref[ok] - 1.0)) < 3e-7
