"""G1: the oracle's tree search (oracle/mz_oracle.c: mzo_uct_search) against the REFERENCE uct_search
(mcts.py:302-407) driven by a scripted network -- bit-exact on every simulation's (parent, action),
the visit counts, the policy vector, the sampled action and the root value."""
import numpy as np
import pytest

from helpers import load_golden

G = load_golden('tree_cases.npz')
N = int(G['num_cases'])


def _case(i):
    return {k[len(f'c{i}_'):]: G[k] for k in G.files if k.startswith(f'c{i}_')}


@pytest.mark.parametrize('i', range(N))
def test_tree_case_bit_exact(oracle, i):
    c = _case(i)
    cfg = oracle.make_config(
        int(c['A']), int(c['sims']), float(c['discount']), bool(c['board']),
        (float(c['kb_min']), float(c['kb_max'])) if int(c['has_bounds']) else None, float(c['alpha']), float(c['eps']),
        float(c['pb_c_base']), float(c['pb_c_init']),
    )
    net = oracle.Net.scripted(c['pi0'], c['values'], c['rewards'])
    r = oracle.uct_search(
        cfg, net, np.zeros(1, np.float32), c['mask'], int(c['cur_player']), int(c['opp_player']), float(c['temperature']),
        bool(c['deterministic']), noise=c['noise'] if int(c['has_noise']) else None, u_tie=c['u_tie'], u_final=float(c['u_final']),
    )
    np.testing.assert_array_equal(r['trace_parent'], c['trace_parent'])
    np.testing.assert_array_equal(r['trace_action'], c['trace_action'])
    np.testing.assert_array_equal(r['visits'], c['visits'])
    assert r['n_tie_used'] == int(c['n_tie'])
    np.testing.assert_array_equal(r['pi'], c['out_pi'])  # float64, bit for bit
    assert r['action'] == int(c['out_action'])
    assert r['root_value'] == float(c['out_root_value'])


def test_tie_stream_exhaustion_is_reported(oracle):
    c = _case(15)  # uniform prior, zero values: a tie at every level
    cfg = oracle.make_config(int(c['A']), int(c['sims']), 1.0, True, (-1, 1), 0.0, 0.25)
    net = oracle.Net.scripted(c['pi0'], c['values'], c['rewards'])
    with pytest.raises(RuntimeError):
        oracle.uct_search(cfg, net, np.zeros(1, np.float32), c['mask'], 1, 2, 1.0, False, u_tie=np.full(3, 0.5))
