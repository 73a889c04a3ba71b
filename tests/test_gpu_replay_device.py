"""SURVEY 8 f1 on the device: replay sampling and priority updates by the kernels of libmzlearner_hip.so (muzero_amd/csrc/learner_replay.hip,
`replay.DeviceSampler`) for a ring whose bookkeeping lives in HBM.  The host path of `PrioritizedReplay` stays the draw-for-draw parity mode
(tests/test_learner.py, recorded reference draws); here the DISTRIBUTION of the device draws is held to the reference's definition (replay.py:81-113):
uniform picks over the live items, proportional picks ~ priority ^ alpha (chi-square), importance weights equal to the host formula on the same picks,
last-write-wins priority updates."""
import numpy as np
import pytest
import torch

from muzero_amd.replay import PrioritizedReplay, Transition

pytestmark = pytest.mark.gpu


def _ring(cap, n_items, alpha, beta, prios):
    dev = torch.device('cuda', 0)
    rp = PrioritizedReplay(cap, alpha, beta, np.random.RandomState(0), device='cuda')
    rp.allocate(dict(state=(3,), action=(2,), pi_prob=(2, 4), value=(2,), reward=(2,)))
    prio_t, count_t = rp.attach_device_writer()  # what Planner.attach_replay does: from here on the device owns counter and priorities
    prio_t[:n_items] = torch.from_numpy(np.asarray(prios, np.float32)).to(dev)
    count_t.fill_(n_items)
    return rp, prio_t, count_t


def _chi2(counts, probs, n):
    exp = probs * n
    keep = exp > 5
    return float(((counts[keep] - exp[keep]) ** 2 / exp[keep]).sum()), int(keep.sum()) - 1


def test_uniform_draws_cover_the_live_items_evenly():
    cap, n_items, B = 4096, 1000, 1 << 16
    rp, _, _ = _ring(cap, n_items, 0.0, 0.0, np.ones(n_items))
    s = rp.device_sampler(seed=5)
    idx, w, ring = s.sample(B)
    assert w is None and ring is rp._ring
    i = idx.cpu().numpy()
    assert i.min() >= 0 and i.max() < n_items  # never a slot that was not written yet
    chi, dof = _chi2(np.bincount(i, minlength=n_items).astype(np.float64), np.full(n_items, 1.0 / n_items), B)
    assert chi < dof + 6 * np.sqrt(2 * dof), (chi, dof)
    i2 = s.sample(B)[0].cpu().numpy()  # the next draw number: a different batch
    assert (i2 != i).mean() > 0.99
    s2 = rp.device_sampler(seed=5)  # same seed, same draw number: the same picks (Philox is keyed, not stateful)
    assert np.array_equal(s2.sample(B)[0].cpu().numpy(), i)


@pytest.mark.parametrize('alpha,beta,n_items,cap', [(0.6, 0.4, 700, 1024), (1.0, 1.0, 5000, 5000), (0.5, 0.5, 3000, 1 << 20)])
def test_proportional_draws_and_importance_weights(alpha, beta, n_items, cap):
    rs = np.random.RandomState(3)
    pr = rs.uniform(0.01, 2.0, n_items).astype(np.float32)
    pr[rs.randint(0, n_items, n_items // 10)] = 0.0  # zero priority: never drawn (replay.py:90-92)
    B = 1 << 17
    rp, prio_t, _ = _ring(cap, n_items, alpha, beta, pr)
    s = rp.device_sampler(seed=11)
    idx, w, _ = s.sample(B)
    i, wv = idx.cpu().numpy(), w.cpu().numpy()
    assert i.min() >= 0 and i.max() < n_items and (pr[i] > 0).all()
    scaled = pr.astype(np.float64) ** alpha
    probs = scaled / scaled.sum()
    chi, dof = _chi2(np.bincount(i, minlength=n_items).astype(np.float64), probs, B)
    assert chi < dof + 6 * np.sqrt(2 * dof), (chi, dof)
    # importance weights: the host formula (replay.py:96-98) on the same picks
    ref = ((1.0 / n_items) / probs[i]) ** beta
    ref /= ref.max()
    np.testing.assert_allclose(wv, ref, rtol=2e-5, atol=1e-7)


def test_priority_updates_last_write_wins_and_feed_the_next_draw():
    n_items, cap = 64, 64
    rp, prio_t, _ = _ring(cap, n_items, 1.0, 0.0, np.full(n_items, 1e-6))
    s = rp.device_sampler(seed=1)
    dev = prio_t.device
    idx = torch.tensor([3, 9, 3, 20, 9, 3], dtype=torch.int64, device=dev)
    val = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0], dtype=torch.float32, device=dev)
    s.update_priorities(idx, val)
    p = prio_t.cpu().numpy()
    assert p[3] == 6.0 and p[9] == 5.0 and p[20] == 4.0 and np.all(np.delete(p, [3, 9, 20]) == np.float32(1e-6))
    picks = s.sample(4096)[0].cpu().numpy()
    assert set(np.unique(picks)) <= {3, 9, 20} | set(range(64)) and np.isin(picks, [3, 9, 20]).mean() > 0.999


def test_hip_learner_trains_from_device_draws_without_host_reads():
    """The `train` leg's loop in miniature: sampler -> HipLearner.step -> sampler.update_priorities, all on device tensors."""
    from helpers import build_mlp, mlp_case
    from muzero_amd.hip_learner import HipLearner

    dev = torch.device('cuda', 0)
    case = mlp_case('cartpole')
    net = build_mlp(case).to(dev)
    cap, n_items, B, K, A = 512, 300, 64, 5, 2
    rp = PrioritizedReplay(cap, 0.6, 0.4, np.random.RandomState(0), device='cuda')
    rp.allocate(dict(state=(4, 5), action=(K,), pi_prob=(K, A), value=(K,), reward=(K,)))
    prio_t, count_t = rp.attach_device_writer()
    g = torch.Generator(device='cpu').manual_seed(1)
    rp._ring['state'][:n_items] = (torch.rand(n_items, 4, 5, generator=g) * 2 - 1).to(dev)
    rp._ring['action'][:n_items] = torch.randint(0, A, (n_items, K), generator=g).to(torch.int8).to(dev)
    rp._ring['pi_prob'][:n_items] = 0.5
    rp._ring['value'][:n_items] = (torch.rand(n_items, K, generator=g) * 20).to(dev)
    rp._ring['reward'][:n_items] = 1.0
    prio_t[:n_items] = 1.0
    count_t.fill_(n_items)
    hl = HipLearner(net, dev, K, B, lr=1e-3)
    s = rp.device_sampler(seed=2)
    ring = dict(rp._ring, state=rp._ring['state'].reshape(cap, -1))
    losses = []
    for _ in range(5):
        idx, w, _ = s.sample(B)
        loss, prio = hl.step(ring, idx, w, B)
        s.update_priorities(idx, prio)
        losses.append(loss.clone())
    torch.cuda.synchronize()
    p = prio_t.cpu().numpy()
    assert np.isfinite(p).all() and (p[:n_items] != 1.0).sum() > B // 2 and (p[n_items:] == 0).all()
    assert all(np.isfinite(float(x)) for x in losses)
