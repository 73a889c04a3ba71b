"""Learner-side rows of SURVEY 8 (f1 replay, f2 learner step) against vectors recorded from the reference
(`oracle/gen_golden.py learn` -> tests/golden/learn_cases.npz): PrioritizedReplay draw-for-draw, the 2-hot target
projection, calc_loss (loss, priorities, every parameter gradient) and three Adam / MultiStepLR updates.
Both sides are float32 torch on CPU with the same op sequence: tolerance 1e-6 relative (BLAS summation order only)."""
import queue
import threading
import types

import numpy as np
import pytest
import torch

from helpers import build_conv, build_mlp, conv_case, load_golden, mlp_case
from muzero_amd import learner
from muzero_amd.replay import PrioritizedReplay, Transition

G = load_golden('learn_cases.npz')
TOL = dict(rtol=2e-5, atol=2e-7)


@pytest.mark.parametrize('j', range(int(G['replay_n'])))
@pytest.mark.parametrize('device', ['cpu'])
def test_replay_matches_reference_draw_for_draw(j, device):
    cap, n_add, pexp, isexp = G[f'replay_{j}_cfg']
    rp = PrioritizedReplay(int(cap), float(pexp), float(isexp), np.random.RandomState(5 + j), device=device)
    items = Transition(*[G[f'replay_{j}_items_{f}'] for f in Transition._fields])
    for i in range(int(n_add)):
        rp.add(Transition(*[x[i] for x in items]), float(G[f'replay_{j}_prios'][i]))
    assert rp.size == int(G[f'replay_{j}_size']) and rp.num_added == int(n_add) and rp.capacity == int(cap)
    np.random.seed(100 + j)
    batch, idx, w = rp.sample(6)
    np.testing.assert_array_equal(idx, G[f'replay_{j}_s1_idx'])
    np.testing.assert_array_equal(w, G[f'replay_{j}_s1_w'])
    for f in Transition._fields:
        got = getattr(batch, f)
        np.testing.assert_array_equal(got, G[f'replay_{j}_s1_{f}'])
        assert got.dtype == G[f'replay_{j}_s1_{f}'].dtype
    rp.update_priorities(idx[:3], [0.5, 1.5, 2.5])
    np.random.seed(200 + j)
    batch2, idx2, w2 = rp.sample(4)
    np.testing.assert_array_equal(idx2, G[f'replay_{j}_s2_idx'])
    np.testing.assert_array_equal(w2, G[f'replay_{j}_s2_w'])
    np.testing.assert_array_equal(batch2.state, G[f'replay_{j}_s2_state'])


def test_replay_errors_and_batch_add():
    with pytest.raises(ValueError):
        PrioritizedReplay(0, 0.0, 0.0, np.random.RandomState(0))
    rp = PrioritizedReplay(8, 0.0, 0.0, np.random.RandomState(0))
    tr = Transition(np.zeros((2, 2), np.float32), np.zeros(5, np.int8), np.full((5, 3), 1 / 3, np.float32), np.zeros(5, np.float32), np.zeros(5, np.float32))
    with pytest.raises(ValueError):
        rp.add(tr, float('nan'))
    with pytest.raises(ValueError):
        rp.add(tr, -1.0)
    rp.add(tr, 1.0)
    with pytest.raises(RuntimeError):
        rp.sample(2)
    with pytest.raises(ValueError):
        rp.update_priorities([0], [float('inf')])
    n = 11  # wraps around the ring
    rp.add_batch(Transition(np.arange(n * 4, dtype=np.float32).reshape(n, 2, 2), np.zeros((n, 5), np.int8), np.zeros((n, 5, 3), np.float32),
                            np.zeros((n, 5), np.float32), np.zeros((n, 5), np.float32)), np.ones(n))
    assert rp.size == 8 and rp.num_added == 12
    assert rp.get([(12 - 1) % 8])[0].state[0, 0] == (n - 1) * 4
    st = rp.get_state()
    rp2 = PrioritizedReplay(8, 0.0, 0.0, np.random.RandomState(0))
    rp2.set_state(st)
    np.testing.assert_array_equal(rp2.sample(4)[0].state, rp.sample(4)[0].state)


@pytest.mark.parametrize('S', [31, 61, 601])
def test_target_projection_matches_reference(S):
    got = learner.scalar_to_categorical_probabilities(torch.from_numpy(G['proj_x']), S).numpy()
    np.testing.assert_allclose(got, G[f'proj_{S}'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(got.sum(-1), 1.0, atol=1e-5)


CASES = [('mlp_cat', 'mlp', 'tiny'), ('mlp_mse', 'mlp', 'tiny_mse'), ('conv_board3', 'conv', 'board3'), ('conv_atari_s', 'conv', 'atari_s')]


def _net(kind, cname):
    net = build_mlp(mlp_case(cname)) if kind == 'mlp' else build_conv(conv_case(cname))
    net.train()
    return net


@pytest.mark.parametrize('name,kind,cname', CASES, ids=[c[0] for c in CASES])
def test_calc_loss_and_updates_match_reference(name, kind, cname):
    pre = f'learn_{name}'
    net = _net(kind, cname)
    big = 1e-5 if cname == 'atari_s' else 0
    tr = Transition(*[G[f'{pre}_{f}'] for f in Transition._fields])
    weights = torch.from_numpy(G[f'{pre}_weights'])
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[2], gamma=0.1)
    losses = []
    for step in range(3):
        opt.zero_grad()
        loss, prio = learner.calc_loss(net, torch.device('cpu'), tr, weights)
        loss.backward()
        if step == 0:
            np.testing.assert_allclose(prio, G[f'{pre}_prio'], **TOL)
            for pn, pp in net.named_parameters():
                ref = G[f'{pre}_grad_{pn}']
                # (the Atari net sums 2304-position planes: float32 rounding of differently blocked sums, relative to the tensor's size)
                np.testing.assert_allclose(pp.grad.numpy(), ref, rtol=2e-4, atol=2e-7 if big == 0 else big * float(np.abs(ref).max()), err_msg=pn)
        if step == 1:
            torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0)
        opt.step()
        sched.step()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, G[f'{pre}_losses'], rtol=2e-5 if big == 0 else 5e-4)
    for pn, pp in net.state_dict().items():
        # (Atari: Adam's first steps are lr * sign(g) -- an entry whose gradient is at rounding distance from zero may step the other way)
        np.testing.assert_allclose(pp.numpy(), G[f'{pre}_final_{pn}'], rtol=2e-4, atol=2e-6 if big == 0 else 1e-3, err_msg=pn)


def test_run_training_loop_with_collector_thread(tmp_path):
    """pipeline.py:170-286 + :491-538: collector thread feeds the replay, the learner runs N steps, writes the reference's
    checkpoint dict, refreshes the actor network, then stops the pipeline."""
    from muzero_amd.config import make_tictactoe_config
    from muzero_amd.pipeline import load_checkpoint

    net, actor = _net('mlp', 'tiny_mse'), _net('mlp', 'tiny_mse')
    cfg = make_tictactoe_config(num_training_steps=6, batch_size=4, min_replay_size=8, use_tensorboard=False)
    cfg.checkpoint_interval = 3
    cfg.train_delay = 0.0
    rs = np.random.RandomState(0)
    q = queue.Queue()
    for _ in range(12):
        q.put((Transition(rs.uniform(-1, 1, (2, 2, 2)).astype(np.float32), rs.randint(0, 5, 5).astype(np.int8),
                          rs.dirichlet(np.ones(5), 5).astype(np.float32), rs.uniform(-1, 1, 5).astype(np.float32),
                          rs.uniform(-1, 1, 5).astype(np.float32)), 1.0))
    rp = PrioritizedReplay(64, 0.0, 0.0, np.random.RandomState(1))
    th = threading.Thread(target=learner.run_data_collector, args=(q, rp, 0, None))
    th.start()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[4], gamma=0.1)
    counter, stop, files = types.SimpleNamespace(value=0), threading.Event(), []
    before = {k: v.clone() for k, v in net.state_dict().items()}
    learner.run_training(cfg, net, opt, sched, torch.device('cpu'), actor, rp, q, counter, str(tmp_path), files, stop, stop_grace_seconds=0.0)
    th.join(timeout=5)
    assert not th.is_alive() and stop.is_set() and counter.value == 6 and len(files) == 2
    assert any(not torch.equal(before[k], v) for k, v in net.state_dict().items())
    for k, v in actor.state_dict().items():  # actor refreshed at step 6
        assert torch.equal(v, net.state_dict()[k])
    ck = load_checkpoint(str(tmp_path / 'train_steps_6_final'), torch.device('cpu'))
    assert set(ck) == {'network', 'optimizer', 'lr_scheduler', 'train_steps'} and ck['train_steps'] == 6


def test_attached_device_writer_owns_cursor_and_priorities():
    """while a device epilogue is attached the host never writes the counter or the whole priority array."""
    from muzero_amd.replay import PrioritizedReplay, Transition

    rp = PrioritizedReplay(8, 1.0, 1.0, np.random.RandomState(0))
    item = Transition(np.zeros((4, 5), np.float32), np.zeros(5, np.int8), np.full((5, 2), 0.5, np.float32), np.zeros(5, np.float32), np.zeros(5, np.float32))
    for i in range(3):
        rp.add(item, float(i + 1))
    prio, count = rp.attach_device_writer()
    assert int(count.item()) == 3 and prio[:3].tolist() == [1.0, 2.0, 3.0]
    # the "device" adds two items and publishes the count
    prio[3] = 9.0
    prio[4] = 8.0
    count.fill_(5)
    rp.update_priorities([1, 1, 4], [5.0, 6.0, 4.0])  # scatters three entries, last value per repeated index; nothing else
    assert prio[:5].tolist() == [1.0, 6.0, 3.0, 9.0, 4.0] and int(count.item()) == 5
    assert rp.num_added == 5 and rp.size == 5
    for call in (lambda: rp.add(item, 1.0), lambda: rp.reset(), lambda: rp.add_batch(Transition(*[np.asarray(x)[None] for x in item]), [1.0])):
        with pytest.raises(RuntimeError):
            call()
    rp.detach_device_writer()
    assert rp.num_added == 5 and rp._prio[:5].tolist() == [1.0, 6.0, 3.0, 9.0, 4.0]
    rp.add(item, 2.5)
    assert rp.num_added == 6
