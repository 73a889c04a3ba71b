"""G2: oracle networks (MLP + conv towers) against reference initial/recurrent inference outputs recorded by
oracle/gen_golden.py with seeded weights.  Float32, tolerance stated per quantity: the oracle's dot products are
k-ordered fmaf chains, torch's CPU kernels use another summation order and libm exp."""
import numpy as np
import pytest

from helpers import CONV_CASES, MLP_CASES, build_conv, build_mlp, load_golden

G = load_golden('net_cases.npz')

HID_TOL = dict(rtol=2e-5, atol=2e-6)   # normalised hidden state in [0, 1]
PI_TOL = dict(rtol=2e-5, atol=1e-7)
# value/reward: util.py:27 cancels in float32 (sqrt(.)/2/eps - 1/2/eps); one ulp of its sqrt is ~3e-5 absolute
VAL_TOL = dict(rtol=2e-4, atol=2e-4)


def _oracle_net(oracle, net, case_kind):
    return oracle.Net.from_module(net, case_kind)


def _check(oracle, onet, prefix, teacher_forcing=True):
    h, r0, pi, v = onet.initial_inference(G[f'{prefix}_obs'])
    np.testing.assert_allclose(h, G[f'{prefix}_init_hidden'].reshape(-1), **HID_TOL)
    np.testing.assert_allclose(pi, G[f'{prefix}_init_pi'], **PI_TOL)
    np.testing.assert_allclose(v, G[f'{prefix}_init_value'], **VAL_TOL)
    assert r0 == 0.0 and float(G[f'{prefix}_init_reward']) == 0.0
    for t, a in enumerate(G[f'{prefix}_actions']):
        # feed the REFERENCE's previous hidden state so every step is compared on identical inputs
        h_in = G[f'{prefix}_init_hidden'] if t == 0 else G[f'{prefix}_rec_hidden'][t - 1]
        h, r, pi, v = onet.recurrent_inference(h_in, int(a))
        np.testing.assert_allclose(h, G[f'{prefix}_rec_hidden'][t].reshape(-1), **HID_TOL)
        np.testing.assert_allclose(r, G[f'{prefix}_rec_reward'][t], **VAL_TOL)
        np.testing.assert_allclose(v, G[f'{prefix}_rec_value'][t], **VAL_TOL)
        np.testing.assert_allclose(pi, G[f'{prefix}_rec_pi'][t], **PI_TOL)


@pytest.mark.parametrize('case', MLP_CASES, ids=[c[0] for c in MLP_CASES])
def test_mlp_inference(oracle, case):
    onet = _oracle_net(oracle, build_mlp(case), 'mlp')
    for j in range(3):
        _check(oracle, onet, f'mlp_{case[0]}_{j}')


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_inference(oracle, case):
    onet = _oracle_net(oracle, build_conv(case), 'conv')
    for j in range(2):
        _check(oracle, onet, f'conv_{case[0]}_{j}')


def test_state_dict_layout_matches_checkpoint_contract():
    """key names/shapes of the shipped CartPole checkpoint (network.py:140-267)."""
    net = build_mlp(MLP_CASES[0])
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    expect = {
        'represent_net.net.0.weight': (512, 20), 'represent_net.net.2.weight': (64, 512),
        'dynamics_net.transition_net.0.weight': (512, 66), 'dynamics_net.transition_net.2.weight': (64, 512),
        'dynamics_net.reward_net.0.weight': (512, 64), 'dynamics_net.reward_net.2.weight': (31, 512),
        'prediction_net.policy_net.0.weight': (512, 64), 'prediction_net.policy_net.2.weight': (2, 512),
        'prediction_net.value_net.0.weight': (512, 64), 'prediction_net.value_net.2.weight': (31, 512),
    }
    for k, s in expect.items():
        assert shapes[k] == s
        assert shapes[k.replace('weight', 'bias')] == (s[0],)
    assert len(shapes) == 20
