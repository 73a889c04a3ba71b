"""Host-side mirror of the reference API (muzero_amd.config / mcts / pipeline / network) against values recorded from
the reference (tests/golden/pipe_cases.npz, net_cases.npz), including its own KATs and its error behaviour."""
import numpy as np
import pytest
import torch

from helpers import CONV_CASES, MLP_CASES, build_conv, build_mlp, load_golden
from muzero_amd import config as cfgmod
from muzero_amd import mcts, pipeline

G = load_golden('pipe_cases.npz')
N = load_golden('net_cases.npz')


def test_config_factories_match_reference():
    fields = [str(f) for f in G['config_fields']]
    for name in ('tictactoe', 'gomoku', 'classic', 'atari'):
        c = getattr(cfgmod, f'make_{name}_config')()
        np.testing.assert_array_equal(np.array([float(getattr(c, f)) for f in fields]), G[f'config_{name}'])
        np.testing.assert_array_equal(np.array(c.lr_milestones, np.float64), G[f'config_{name}_milestones'])
        flags = [c.clip_grad, c.use_tensorboard, c.is_board_game, c.known_bounds is not None]
        np.testing.assert_array_equal(np.array(flags, np.int32), G[f'config_{name}_flags'])


def test_temperature_schedules_match_reference():
    rows = []
    for fn in ('tictactoe', 'gomoku', 'classic', 'atari'):
        f = getattr(cfgmod, f'{fn}_visit_softmax_temperature_fn')
        rows.append([f(es, ts) for es in (0, 5, 6, 29, 30, 100) for ts in (0, 29999, 30000, 60000, 499999, 500000, 1000000)])
    np.testing.assert_array_equal(np.array(rows), G['temperature_table'])


def test_n_step_target_kats_and_cases():
    # the reference's own KATs, tests/pipeline_test.py:24-53
    out = pipeline.compute_n_step_target([1.0] * 5, [0] * 5, 5, 0.997)
    np.testing.assert_almost_equal(np.array(out), np.array([4.97, 3.982, 2.991, 1.997, 1.0]), decimal=3)
    np.testing.assert_array_equal(np.array(out), G['nstep_kat1_out'])
    np.testing.assert_array_equal(np.array(pipeline.compute_n_step_target([1.0] * 10, list(G['nstep_kat2_roots']), 5, 0.997)), G['nstep_kat2_out'])
    for j in range(int(G['nstep_n'])):
        out = pipeline.compute_n_step_target(list(G[f'nstep_{j}_rewards']), list(G[f'nstep_{j}_roots']), int(G[f'nstep_{j}_td']),
                                             float(G[f'nstep_{j}_discount']))
        np.testing.assert_array_equal(np.array(out), G[f'nstep_{j}_out'])
    with pytest.raises(ValueError):
        pipeline.compute_n_step_target([1.0], [1.0, 2.0], 5, 0.9)


def test_mc_return_target():
    for j in range(int(G['mc_n'])):
        out = pipeline.compute_mc_return_target(list(G[f'mc_{j}_rewards']), list(G[f'mc_{j}_players']))
        np.testing.assert_array_equal(np.array(out), G[f'mc_{j}_out'])
    with pytest.raises(ValueError):
        pipeline.compute_mc_return_target([1.0], [1, 2])


def test_make_unroll_sequence():
    for j in range(int(G['unroll_n'])):
        obs, acts = list(G[f'unroll_{j}_obs']), [int(a) for a in G[f'unroll_{j}_actions']]
        rews, pis = [float(r) for r in G[f'unroll_{j}_rewards']], list(G[f'unroll_{j}_pis'])
        vals, prios = [float(v) for v in G[f'unroll_{j}_values']], G[f'unroll_{j}_prios']
        seq = list(pipeline.make_unroll_sequence(obs, acts, rews, pis, vals, prios, 5))
        assert len(acts) == len(obs)  # caller's lists are not mutated (unlike pipeline.py:739-747)
        np.testing.assert_array_equal(np.stack([t.state for t, _ in seq]), G[f'unroll_{j}_out_state'])
        np.testing.assert_array_equal(np.stack([t.action for t, _ in seq]), G[f'unroll_{j}_out_action'])
        assert seq[0][0].action.dtype == np.int8
        np.testing.assert_array_equal(np.stack([t.reward for t, _ in seq]), G[f'unroll_{j}_out_reward'])
        np.testing.assert_array_equal(np.stack([t.value for t, _ in seq]), G[f'unroll_{j}_out_value'])
        np.testing.assert_array_equal(np.stack([t.pi_prob for t, _ in seq]), G[f'unroll_{j}_out_pi'])
        np.testing.assert_array_equal(np.array([p for _, p in seq]), G[f'unroll_{j}_out_prio'])
    # action spaces beyond int8 (Gomoku 15x15, A = 226) are representable here
    seq = list(pipeline.make_unroll_sequence([np.zeros(1)] * 2, [200, 225], [0.0, 1.0], [np.full(226, 1 / 226)] * 2, [0.0, 0.0], [0.0, 0.0], 5))
    assert seq[0][0].action.dtype == np.int16 and seq[0][0].action[0] == 200


def test_play_policy_noise_and_mask_helpers():
    for j in range(int(G['policy_n'])):
        T = float(G[f'policy_{j}_T'])
        out = mcts.generate_play_policy(G[f'policy_{j}_visits'], T)
        np.testing.assert_array_equal(out, G[f'policy_{j}_out'])  # same numpy expression as the reference
    for j in range(int(G['noise_n'])):
        noised = mcts.add_dirichlet_noise(G[f'noise_{j}_p'], eps=0.25, alpha=0.25, noise=G[f'noise_{j}_noise'])
        np.testing.assert_array_equal(noised, G[f'noise_{j}_noised'])
        assert noised.dtype == np.float64
        masked = mcts.set_illegal_action_probs_to_zero(G[f'noise_{j}_mask'].astype(bool), noised)
        np.testing.assert_array_equal(masked, G[f'noise_{j}_masked'])
        m32 = mcts.set_illegal_action_probs_to_zero(G[f'noise_{j}_mask'].astype(bool), G[f'noise_{j}_p'])
        assert m32.dtype == np.float32
        np.testing.assert_array_equal(m32, G[f'noise_{j}_masked32'])


def test_error_behaviour_matches_reference():
    with pytest.raises(ValueError):
        mcts.generate_play_policy(np.zeros((2, 2)), 1.0)  # mcts.py:266-267
    with pytest.raises(ValueError):
        mcts.generate_play_policy(np.ones(3), 2.0)  # mcts.py:268-269
    with pytest.raises(ValueError):
        mcts.generate_play_policy(np.ones(3), 1)  # temperature must be float
    with pytest.raises(ValueError):
        mcts.add_dirichlet_noise([0.5, 0.5])  # mcts.py:237-238
    with pytest.raises(ValueError):
        mcts.add_dirichlet_noise(np.ones(2, np.float32) / 2, eps=1.5)
    with pytest.raises(ValueError):
        mcts.add_dirichlet_noise(np.ones(2, np.float32) / 2, alpha=2.0)
    mm = mcts.MinMaxStats(cfgmod.KnownBounds(-1, 1))
    assert mm.normalize(0.0) == 0.5
    mm2 = mcts.MinMaxStats(None)
    assert mm2.normalize(3.0) == 3.0
    mm2.update(1.0); mm2.update(3.0)
    assert mm2.normalize(2.0) == 0.5


def test_uct_search_requires_the_gpu_planner():
    net = build_mlp(MLP_CASES[3])
    c = cfgmod.make_classic_config(use_tensorboard=False)
    with pytest.raises(Exception) as ei:
        mcts.uct_search(np.zeros((3, 4), np.float32), net, torch.device('cpu'), c, 1.0, np.ones(3, bool), 1, 1)
    assert 'no CPU fallback' in str(ei.value) or 'HIP' in str(ei.value)


def _torch_infer(net, obs, actions):
    """The learner-side tensor API (represent / dynamics / prediction) composed like network.py:62-111."""
    from muzero_amd.network import logits_to_transformed_expected_value as l2v

    with torch.no_grad():
        h = net.represent(torch.from_numpy(obs)[None].float())
        pl, v = net.prediction(h)
        out = [(h[0].numpy(), torch.softmax(pl, 1)[0].numpy(), (v if net.mse_loss_for_value else l2v(v, net.value_support_size)).item())]
        for a in actions:
            h, r = net.dynamics(h, torch.tensor([[int(a)]]))
            pl, v = net.prediction(h)
            out.append((h[0].numpy(), torch.softmax(pl, 1)[0].numpy(), (v if net.mse_loss_for_value else l2v(v, net.value_support_size)).item(),
                        (r if net.mse_loss_for_reward else l2v(r, net.reward_support_size)).item()))
    return out


@pytest.mark.parametrize('case', MLP_CASES[:4] + CONV_CASES[:2] + CONV_CASES[3:4], ids=lambda c: c[0])
def test_torch_modules_reproduce_reference_outputs(case):
    """Same state_dict -> same outputs as the reference modules (incl. the scrambled conv action planes)."""
    mlp = len(case) == 8
    net = build_mlp(case) if mlp else build_conv(case)
    pre = f"{'mlp' if mlp else 'conv'}_{case[0]}_0"
    out = _torch_infer(net, N[f'{pre}_obs'], N[f'{pre}_actions'])
    np.testing.assert_allclose(out[0][0], N[f'{pre}_init_hidden'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out[0][1], N[f'{pre}_init_pi'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out[0][2], N[f'{pre}_init_value'], rtol=2e-4, atol=2e-4)
    for t in range(len(N[f'{pre}_actions'])):
        np.testing.assert_allclose(out[t + 1][0], N[f'{pre}_rec_hidden'][t], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(out[t + 1][1], N[f'{pre}_rec_pi'][t], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(out[t + 1][2], N[f'{pre}_rec_value'][t], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(out[t + 1][3], N[f'{pre}_rec_reward'][t], rtol=2e-4, atol=2e-4)


def test_checkpoint_layout_round_trip(tmp_path):
    """{'network','optimizer','lr_scheduler','train_steps'} (pipeline.py:224-230) written and read back."""
    net = build_mlp(MLP_CASES[3])
    opt = torch.optim.Adam(net.parameters(), lr=0.005, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[10], gamma=0.1)
    f = str(tmp_path / 'CartPole-v1_train_steps_10')
    pipeline.create_checkpoint({'network': net.state_dict(), 'optimizer': opt.state_dict(), 'lr_scheduler': sched.state_dict(), 'train_steps': 10}, f)
    ck = pipeline.load_checkpoint(f, 'cpu')
    assert sorted(ck.keys()) == ['lr_scheduler', 'network', 'optimizer', 'train_steps']
    net2 = build_mlp(MLP_CASES[3])
    net2.load_state_dict(ck['network'])
    assert list(ck['network'].keys()) == list(net.state_dict().keys())


def test_episode_assembler_matches_reference_episode():
    """Device-style record stream -> (Transition, priority) items == the reference's run_self_play output (fixture G6)."""
    P = load_golden('selfplay_cases.npz')
    c = cfgmod.make_tictactoe_config(use_tensorboard=False)
    for ep in range(int(P['n_episodes'])):
        n = int(P[f'ep{ep}_n_moves'])
        rec = dict(
            obs=P[f'ep{ep}_search_obs'][:, None].astype(np.float32), action=P[f'ep{ep}_search_action'][:, None],
            pi=P[f'ep{ep}_search_pi'][:, None], root_value=P[f'ep{ep}_search_root'][:, None], player=P[f'ep{ep}_search_cur'][:, None],
            reward=P[f'ep{ep}_tr_reward'][:, :1].astype(np.float32), done=np.zeros((n, 1), np.uint8),
        )
        rec['done'][-1, 0] = 1
        items = list(pipeline.EpisodeAssembler(c, 1).feed(rec))
        assert len(items) == n
        np.testing.assert_array_equal(np.stack([t.state for t, _ in items]), P[f'ep{ep}_tr_state'].astype(np.float32))
        np.testing.assert_array_equal(np.stack([t.action for t, _ in items]), P[f'ep{ep}_tr_action'])
        np.testing.assert_array_equal(np.stack([t.reward for t, _ in items]), P[f'ep{ep}_tr_reward'])
        np.testing.assert_array_equal(np.stack([t.value for t, _ in items]), P[f'ep{ep}_tr_value'])
        np.testing.assert_array_equal(np.stack([t.pi_prob for t, _ in items]), P[f'ep{ep}_tr_pi'])
        np.testing.assert_array_equal(np.array([p for _, p in items]), P[f'ep{ep}_tr_priority'])


def test_network_modules_pickle_and_deepcopy_without_their_engine():
    """The reference's launchers hand `actor_network` to spawned actor processes (classic/run_training.py:100,168-186): a module whose HIP
    inference engine is already bound (a ctypes handle) must still pickle and deep-copy -- the copy binds its own engine on first use."""
    import copy
    import pickle

    import torch

    net = build_mlp(MLP_CASES[0])
    net._engine, net._engine_version = object(), ('stale',)  # (stand-in for a bound engine: no GPU here)
    for clone in (copy.deepcopy(net), pickle.loads(pickle.dumps(net))):
        assert clone._engine is None and clone._engine_version is None
        for (k, x), (_, y) in zip(net.state_dict().items(), clone.state_dict().items()):
            assert torch.equal(x, y), k
    assert net._engine is not None  # (the original keeps its engine)
