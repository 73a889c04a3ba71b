"""-m gpu: the HIP learner on MuZeroAtariNet (muzero_amd/csrc/mz_learn_conv.h, the tile path of DESIGN.md 4c) -- the reference's update
(pipeline.py:541-629 `calc_loss`, pipeline.py:238-255 backward / clip / Adam / MultiStepLR in `run_training`; network.py:312-353 the Atari
representation, network.py:501-537 `MuZeroAtariNet`) for 96 x 96 frame stacks: strided
convolutions, 12 x 12 tiles of the 48 x 48 and 24 x 24 stages, average pools, categorical (2-hot cross-entropy) value / reward heads.

Checked against (1) the committed fixture the reference itself produced (`oracle/gen_golden.py learn` -> learn_conv_atari_s_*: loss, priorities,
every gradient tensor, three updates) and (2) PyTorch float64 autograd of this repo's network module on seeded batches of other shapes.

Tolerances (round 6).  GRADIENTS of this network sit on ReLU kinks on practically every batch: one layer of the 48 x 48 stage applies ReLU to
128 x 2304 values per frame stack, some pre-activation lies within float32 rounding of zero, a float32 pass -- this one, PyTorch-ROCm's or the
reference's own -- takes the other branch there, and every upstream tensor moves by that element's contribution (up to 1e-1 of a tensor's largest
entry).  Rounds 4-5 answered with loose bars (0.25 on random weights).  Now the float64 reference is TOLD what the HIP forward pass decided --
every ReLU mask and every arg-min / arg-max of normalize_hidden_state, read back through the library's diagnostic hook (tests/forced_masks.py) --
and takes the same branches; a float32 and a float64 pass are then compared on one piecewise-linear function, where an error of the tile path
(halo positions, 12 x 16 tiles, parity planes, pooling) is a full-size error and rounding is 1e-5:
  * seeded RANDOM weights, every shape: every gradient tensor within 5e-4 of its largest entry (8 of the 10 committed shapes within 1e-4: measured 2e-5; the 3-image shapes 3e-4) -- at the Atari config's full size
    (128 planes, 8 blocks, batch 128) within 4 x what PyTorch-ROCm's own float32 autograd reaches on the same branch (4.6e-4 there; the kernels 6.2e-4);
  * the reference's own fixture batch: the kernels against float64 on their own branch at 4 x PyTorch-ROCm's float32 (5e-4 .. 1.7e-3 on this 2-image batch) on EVERY network part -- and against the fixture's
    gradients (the reference's float32 run, which took ITS branches) at 8e-2 on the representation net, 3e-3 elsewhere, as before;
  * KINK-FREE weights (`kinkfree_state_dict`) remain as an independent second check that involves no read-back of the library's tensors: every ReLU
    channel on or off for the whole batch, plain float64 autograd, 3e-3 (measured 1e-5).
Loss, priorities and BatchNorm statistics are continuous in the weights and held to the board-net bars everywhere."""
import copy

import numpy as np
import pytest
import torch

from helpers import build_conv, conv_case, load_golden, seeded_state_dict
from muzero_amd import learner
from muzero_amd.replay import Transition

pytestmark = pytest.mark.gpu
G = load_golden('learn_cases.npz')
REP_TOL, TIGHT = 8e-2, 3e-3  # (REP_TOL: against the REFERENCE's float32 run, which took its own ReLU branches; TIGHT: kink-free weights, typical agreement 1e-5)


def _hip(net, dev, max_batch, K=5, **kw):
    from muzero_amd.hip_learner import HipLearner

    kw.setdefault('lr', 1e-3)
    return HipLearner(net, dev, K, max_batch, **kw)


def _ring(tr, dev):
    B = tr.state.shape[0]
    return dict(state=torch.from_numpy(tr.state).to(dev).reshape(B, -1).contiguous(), action=torch.from_numpy(tr.action).to(dev),
                pi_prob=torch.from_numpy(tr.pi_prob).to(dev), value=torch.from_numpy(tr.value).to(dev), reward=torch.from_numpy(tr.reward).to(dev))


def _rel(a, ref):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(ref, np.float64)).max()) / max(1e-8, float(np.abs(ref).max()))


def kinkfree_state_dict(net, seed):
    """seeded_state_dict with every ReLU pushed off its kink: BatchNorm weight in [0.08, 0.15], |bias| in [2, 3] (positive in the second
    conv of a residual block, whose sum with the non-negative skip is then positive; a random sign per channel elsewhere: dead and live channels
    mixed), and one sign per output channel for conv_1 / conv_2, whose ReLUs act on the convolution of a non-negative input directly
    (network.py:151-152)."""
    sd = seeded_state_dict(net, seed)
    rs = np.random.RandomState(seed + 77)
    for name, m in net.named_modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            n = m.num_features
            sign = np.ones(n) if name.endswith('conv_block2.1') else rs.choice([-1.0, 1.0, 1.0], size=n)
            sd[name + '.weight'] = torch.from_numpy(rs.uniform(0.08, 0.15, n).astype(np.float32))
            sd[name + '.bias'] = torch.from_numpy((sign * rs.uniform(2.0, 3.0, n)).astype(np.float32))
    for name in ('represent_net.conv_1.weight', 'represent_net.conv_2.weight'):
        if name not in sd:  # (board nets: every ReLU follows a BatchNorm)
            continue
        wt = sd[name].abs()
        sign = torch.from_numpy(rs.choice([-1.0, 1.0, 1.0], size=wt.shape[0]).astype(np.float32))
        sd[name] = wt * sign.view(-1, 1, 1, 1)
    return sd


def test_loss_gradients_and_three_updates_match_the_reference():
    """gen_golden.gen_learn's recipe on the reference's MuZeroAtariNet (4 x 96 x 96 frames, 8 planes, 1 block, supports 11): Adam(lr 1e-3),
    MultiStepLR([2], 0.1), clip_grad_norm_(10) on the second step only, the network in train mode."""
    pre = 'learn_conv_atari_s'
    dev = torch.device('cuda', 0)
    net = build_conv(conv_case('atari_s')).to(dev)
    net.train()
    hl = _hip(net, dev, 8, lr=1e-3, milestones=[2], gamma=0.1, max_grad_norm=10.0)
    assert hl.kind == 'atari'
    tr = Transition(*[G[f'{pre}_{f}'] for f in Transition._fields])
    B = tr.state.shape[0]
    ring = _ring(tr, dev)
    w = torch.from_numpy(G[f'{pre}_weights']).to(dev)
    losses, worst = [], {}
    for step in range(3):
        loss, prio = hl.grad(ring, None, w, B)
        if step == 0:
            np.testing.assert_allclose(prio.cpu().numpy(), G[f'{pre}_prio'], rtol=1e-3, atol=1e-3)
            for pn in hl.views:
                e = _rel(hl.grad_views[pn].cpu().numpy(), G[f'{pre}_grad_{pn}'])
                net_name = pn.split('.')[0]
                worst[net_name] = max(worst.get(net_name, 0.0), e)
                assert e <= (REP_TOL if net_name == 'represent_net' else TIGHT), (pn, e)
            # ... and the same batch against float64 autograd on THIS pass's branches (tests/forced_masks.py): every part of the network, 1e-4
            from test_gpu_conv_learner import _same_branch, same_branch_bar

            errs, err32, _, _, flipped = _same_branch(hl, net, tr, G[f'{pre}_weights'], B, 5, dev)
            k, e, bar = same_branch_bar(errs, err32)
            assert e <= bar and bar <= 3e-3, (k, e, bar, flipped)  # (this batch of 2 frame stacks: PyTorch-ROCm float32 4.2e-4 on the same branch, the kernels 5.2e-4)
        hl.apply(clip=(step == 1))
        losses.append(float(loss))
    print('worst relative gradient difference per network:', worst)
    # (later losses: Adam's first steps are lr * sign(g), and most weights of a layer have |g| far below the layer's largest entry, i.e. below the
    # mask noise on it -- their steps go either way, in this run as in the reference's; measured 5e-4 and 1.2e-2 here)
    np.testing.assert_allclose(losses, G[f'{pre}_losses'], rtol=3e-2)
    np.testing.assert_allclose(losses[0], G[f'{pre}_losses'][0], rtol=1e-5)
    sd = net.state_dict()
    # After the first update the two runs no longer hold the same weights: Adam's first step is lr * sign(g) whatever |g|, so where the mask noise
    # above outweighs a near-zero gradient entry the weight moves the other way (by at most lr per step), and the activations behind it shift.
    # The bars below are what that allows; the first step's gradients, statistics and loss are held to the tight bars above and in the sweep.
    moved = 1e-3 + 1e-3 + 1e-4
    bad = []
    for pn in sd:
        ref = G[f'{pre}_final_{pn}']
        got = sd[pn].cpu().numpy()
        if 'num_batches_tracked' in pn:
            assert int(got) == int(ref), pn
            continue
        d = np.abs(got - ref)
        if 'running' in pn:
            ok = float(d.max()) <= 2e-2 * max(1.0, float(np.abs(ref).max()))
        else:
            ok = float(d.max()) <= 2 * moved + 1e-6 and float(np.median(d)) <= 2e-4
        if not ok:
            bad.append((pn, float(d.max()), float(np.median(d))))
    assert not bad, bad
    assert abs(hl.current_lr() - 1e-4) < 1e-12 and hl.steps == 3


class _ReluProbe:
    """While active, records the smallest non-zero |pre-activation| of the float64 pass over (a) every ReLU and (b) the ReLUs of the 6 x 6
    dynamics / prediction towers and the heads' MLPs only (inputs of at most 6 x 6 positions)."""

    def __init__(self):
        self.closest_all = self.closest_small = self.closest_tie = float('inf')

    def __enter__(self):
        import torch.nn.functional as F

        from muzero_amd import network as nw

        self._F, self._relu, self._nw, self._norm = F, F.relu, nw, nw.normalize_hidden_state
        probe = self

        def normalize(hs):  # the one kink kink-free weights do not remove: two channels of a position tie for its maximum / minimum (util.py:31-36),
            v = hs.detach().flatten(2) if hs.dim() > 2 else hs.detach()  # relative to the values' size (what float32 can resolve)
            top = v.topk(2, dim=1).values
            low = (-v).topk(2, dim=1).values
            scale = v.abs().amax(dim=1).clamp_min(1e-30)
            probe.closest_tie = min(probe.closest_tie, float(((top[:, 0] - top[:, 1]) / scale).min()), float(((low[:, 0] - low[:, 1]).abs() / scale).min()))
            return probe._norm(hs)

        nw.normalize_hidden_state = normalize

        def relu(x, inplace=False):
            nz = x.detach().abs()
            nz = nz[nz > 0]
            if nz.numel():
                m = float(nz.min())
                probe.closest_all = min(probe.closest_all, m)
                if x.dim() < 4 or x.shape[-1] <= 6:
                    probe.closest_small = min(probe.closest_small, m)
            return probe._relu(x, inplace=False)

        F.relu = relu
        return self

    def __exit__(self, *exc):
        self._F.relu = self._relu
        self._nw.normalize_hidden_state = self._norm


def _f64(net, tr, w, dev):
    net_d = copy.deepcopy(net).double()
    net_d.train()
    t = lambda x, dt: torch.from_numpy(np.asarray(x)).to(dev).to(dt)  # noqa: E731
    with _ReluProbe() as probe:
        loss, prio = learner.loss_tensors(net_d, t(tr.state, torch.float64), t(tr.action, torch.int64), t(tr.value, torch.float64), t(tr.reward, torch.float64),
                                          t(tr.pi_prob, torch.float64), t(w, torch.float64))
    loss.backward()
    return float(loss.detach()), prio.detach(), {k: p.grad for k, p in net_d.named_parameters()}, net_d.state_dict(), probe


# frames, planes, blocks, actions, value support, reward support, batch, unroll steps, seed
SHAPES = [(4, 8, 1, 6, 11, 11, 3, 5, 3), (4, 8, 1, 6, 11, 11, 3, 5, 5), (4, 8, 1, 6, 11, 11, 3, 5, 7), (4, 8, 1, 6, 11, 11, 3, 5, 2),
          (4, 16, 2, 18, 61, 31, 5, 5, 1), (32, 24, 1, 4, 601, 601, 2, 2, 2), (4, 128, 1, 6, 61, 61, 2, 3, 3), (1, 8, 3, 3, 5, 7, 1, 1, 4),
          (2, 40, 1, 9, 21, 21, 4, 4, 6), (4, 8, 1, 6, 11, 11, 9, 5, 9)]
SAME_BRANCH = {}
ATARI_FLAT = 5e-4  # random weights, float64 on the HIP pass's own branches: every tensor within this of its largest entry (rounds 4-5: 0.25)


def _case(chan, planes, blocks, A, vs, rs_, B, K, seed, kinkfree, dev, int8_state=False):
    from muzero_amd.network import MuZeroAtariNet

    net = MuZeroAtariNet((chan, 96, 96), A, blocks, planes, vs, rs_)
    net.load_state_dict(kinkfree_state_dict(net, 100 + seed) if kinkfree else seeded_state_dict(net, 100 + seed))
    net = net.to(dev)
    net.train()
    rs = np.random.RandomState(seed)
    tr = Transition(rs.uniform(0, 1, (B, chan, 96, 96)).astype(np.float32), rs.randint(0, A, (B, K)).astype(np.int8),
                    rs.dirichlet(np.ones(A), size=(B, K)).astype(np.float32), (rs.uniform(-1, 1, (B, K)) * 8.0).astype(np.float32),
                    rs.uniform(-1, 1, (B, K)).astype(np.float32))
    if int8_state:  # (the replay ring's int8 state storage: the gather kernel converts)
        tr = tr._replace(state=(tr.state * 4.0).astype(np.int8))
    w = rs.uniform(0.3, 1.0, B).astype(np.float32)
    loss_d, prio_d, gd, sd_d, probe = _f64(net, tr._replace(state=tr.state.astype(np.float64)), w, dev)
    probe.err32 = f32_errors(net, tr, w, dev, gd) if kinkfree else None
    hl = _hip(net, dev, B, K=K)
    loss, prio = hl.grad(_ring(tr, dev), None, torch.from_numpy(w).to(dev), B)
    assert abs(float(loss) - loss_d) <= 1e-4 * max(1.0, abs(loss_d))
    np.testing.assert_allclose(prio.cpu().numpy(), prio_d.cpu().numpy(), rtol=1e-3, atol=2e-4 * max(1.0, float(prio_d.abs().max())))
    sd = net.state_dict()  # the train-mode pass has updated the running statistics, one momentum step per application of a layer
    # (the kernels sum the batch statistics around a pivot -- DESIGN 4c: the variance keeps its accuracy whatever mean^2 / var is; the kink-free
    # weights drive the heads to mean^2 / var ~ 3 500, where E[y^2] - mean^2 from float32 partial sums was off by 3e-4)
    stat_tol = 3e-5
    for k, v in sd_d.items():
        if 'running' in k:
            assert float((v - sd[k].double()).abs().max()) <= stat_tol * max(1.0, float(v.abs().max())), k
        if 'num_batches_tracked' in k:
            assert int(v) == int(sd[k]), k
    return grad_errors(gd, hl.grad_views), probe


def f32_errors(net, tr, w, dev, gd):
    """grad_errors of PyTorch-ROCm's own float32 autograd on the same batch: how well float32 CAN do on it.  Kink-free weights remove the ReLU
    kinks, not ill-conditioning: a BatchNorm channel whose batch variance is tiny (a plane that is almost constant: dead input channels, one
    frame stack) multiplies the rounding of y - mean by up to 1 / sqrt(1e-5) = 316, in any float32 implementation."""
    n32 = copy.deepcopy(net)
    n32.train()
    t = lambda x, dt: torch.from_numpy(np.asarray(x)).to(dev).to(dt)  # noqa: E731
    loss, _ = learner.loss_tensors(n32, t(tr.state, torch.float32), t(tr.action, torch.int64), t(tr.value, torch.float32), t(tr.reward, torch.float32),
                                   t(tr.pi_prob, torch.float32), t(w, torch.float32))
    loss.backward()
    return grad_errors(gd, {k: p.grad for k, p in n32.named_parameters()})


def kinkfree_worst(errs, err32, tie=float('inf')):
    """The kink-free bar: every tensor within TIGHT of float64 autograd -- or, on an ill-conditioned batch, within 4 x what PyTorch-ROCm's float32
    autograd manages on that tensor (never beyond 8e-2); `tie` < 1e-6 (two channels of a position within float32 resolution of each other at a
    normalisation's minimum / maximum: the gradient goes to ONE of them, and which one is a coin toss in float32 -- once in ~900 fuzz cases): the
    mask-noise bar.  Returns (tensor, error, bar) of the worst offender relative to its bar."""
    worst = None
    for k, e in errs.items():
        bar = 0.25 if tie < 1e-6 else min(8e-2, max(TIGHT, 4.0 * (err32[k] if err32 else 0.0)))
        if worst is None or e / bar > worst[1] / worst[2]:
            worst = (k, e, bar)
    return worst


def grad_errors(gd, views):
    """Per tensor: max |difference| relative to the tensor's largest entry, but no finer than 1e-3 of the largest gradient entry of the whole
    network; BatchNorm shifts (sums of dz over every position: the most cancellation-prone tensors) count at a fifth.  With every channel live,
    the shift of a tower's LAST BatchNorm is cancelled by the batch statistics of the layers behind it -- exactly (float64 gradient ~1e-17 of the
    others) or up to border effects of the next convolution's zero padding -- and what a float32 pass leaves there is the rounding of that
    cancelling sum."""
    gmax = max(float(g.abs().max()) for g in gd.values())
    errs = {}
    for k, g in gd.items():
        e = float((g - views[k].double()).abs().max()) / max(float(g.abs().max()), 1e-3 * gmax)
        errs[k] = e / 5 if k.endswith('.1.bias') else e
    return errs


IDS = [f'c{s[0]}-p{s[1]}-r{s[2]}-a{s[3]}-v{s[4]}-n{s[6]}-k{s[7]}-s{s[8]}' for s in SHAPES]


@pytest.mark.parametrize('chan,planes,blocks,A,vs,rs_,B,K,seed', SHAPES, ids=IDS)
def test_gradient_matches_float64_autograd_kink_free(chan, planes, blocks, A, vs, rs_, B, K, seed):
    errs, probe = _case(chan, planes, blocks, A, vs, rs_, B, K, seed, True, torch.device('cuda', 0))
    assert probe.closest_all > 1e-4, ('the construction left a pre-activation near zero', probe.closest_all)
    k, e, bar = kinkfree_worst(errs, probe.err32)  # (the committed shapes get no tie allowance: one of them has a tie at 2.6e-7 and stays inside the tight bar)
    assert e <= bar, (k, e, bar)
    assert any(e > 0 for e in errs.values())


def _random_case(chan, planes, blocks, A, vs, rs_, B, K, seed, dev, int8_state=False):
    from muzero_amd.network import MuZeroAtariNet
    from test_gpu_conv_learner import _same_branch

    net = MuZeroAtariNet((chan, 96, 96), A, blocks, planes, vs, rs_)
    net.load_state_dict(seeded_state_dict(net, 100 + seed))
    net = net.to(dev)
    net.train()
    rs = np.random.RandomState(seed)
    tr = Transition(rs.uniform(0, 1, (B, chan, 96, 96)).astype(np.float32), rs.randint(0, A, (B, K)).astype(np.int8),
                    rs.dirichlet(np.ones(A), size=(B, K)).astype(np.float32), (rs.uniform(-1, 1, (B, K)) * 8.0).astype(np.float32),
                    rs.uniform(-1, 1, (B, K)).astype(np.float32))
    if int8_state:  # (the replay ring's int8 state storage: the gather kernel converts)
        tr = tr._replace(state=(tr.state * 4.0).astype(np.int8))
    w = rs.uniform(0.3, 1.0, B).astype(np.float32)
    hl = _hip(net, dev, B, K=K)
    loss, prio = hl.grad(_ring(tr, dev), None, torch.from_numpy(w).to(dev), B)
    errs, err32, loss_d, prio_d, flipped = _same_branch(hl, net, tr, w, B, K, dev)
    assert abs(float(loss) - loss_d) <= 3e-5 * max(1.0, abs(loss_d))
    np.testing.assert_allclose(prio.cpu().numpy(), prio_d.cpu().numpy(), rtol=1e-3, atol=2e-4 * max(1.0, float(prio_d.abs().max())))
    return errs, err32, flipped


@pytest.mark.parametrize('chan,planes,blocks,A,vs,rs_,B,K,seed', SHAPES, ids=IDS)
def test_gradient_matches_float64_autograd_random_weights(chan, planes, blocks, A, vs, rs_, B, K, seed):
    """Seeded random weights, full-strength element-wise masks in the strided, tiled and pooled stages, against float64 autograd on the HIP pass's
    own branches: every tensor within 1e-4 of its largest entry (the 0.25 bar is gone)."""
    from test_gpu_conv_learner import same_branch_bar

    errs, err32, flipped = _random_case(chan, planes, blocks, A, vs, rs_, B, K, seed, torch.device('cuda', 0))
    # 5e-4 flat for this net (the 3-image shapes sit at 2.5e-4 .. 3.4e-4 on a BatchNorm scale / shift of the 48 x 48 stage: sums of 7 000 signed terms in
    # float32; PyTorch-ROCm's own float32 on the same branch lands between 4e-5 and 1.5e-4 there from run to run -- its convolution backward is not
    # deterministic -- so the yardstick clause alone would flake); most shapes must still meet 1e-4 (next test)
    k, e, bar = same_branch_bar(errs, err32, flat=ATARI_FLAT)
    assert e <= bar and bar <= 3e-3, (k, e, bar, flipped)
    SAME_BRANCH[(chan, planes, blocks, B, seed)] = (e, bar)


def test_most_random_weight_shapes_meet_the_flat_bar():
    """The 5e-4 flat bar and the yardstick clause must stay the exception: most committed shapes agree to 1e-4 (8 of 10 on the final build; six required)."""
    if len(SAME_BRANCH) < len(SHAPES):
        pytest.skip('runs after the parametrised cases')
    print({k: (f'{v[0]:.1e}', f'{v[1]:.1e}') for k, v in SAME_BRANCH.items()})
    assert sum(1 for e, _ in SAME_BRANCH.values() if e <= 1e-4) >= 6, SAME_BRANCH


def test_full_size_atari_config_at_batch_128_random_weights():
    """make_atari_config's network (128 planes, 8 blocks, supports 61, unroll 5) at ITS batch size with seeded RANDOM weights: the NPT = 16 tile builds,
    the launch-time-aware tilings and chunk counts of batch 128 under element-wise masks."""
    from test_gpu_conv_learner import same_branch_bar

    errs, err32, flipped = _random_case(4, 128, 8, 6, 61, 61, 128, 5, 11, torch.device('cuda', 0))
    k, e, bar = same_branch_bar(errs, err32)
    print('Atari config net, batch 128, random weights: worst tensor', k, '%.2e' % e, 'bar %.2e' % bar, '(PyTorch-ROCm float32 on the same branch: %.2e);' % max(err32.values()),
          flipped, 'ReLU decisions differ from float64\'s own')
    assert e <= bar and bar <= 5e-3, (k, e, bar)
    assert flipped > 0


def test_ring_by_index_and_repeat_are_bit_identical():
    """Rows gathered by replay index give the bits of the same rows passed directly; a second pass over the same batch gives the same bits
    (no atomics, fixed reduction orders)."""
    dev = torch.device('cuda', 0)
    net = build_conv(conv_case('atari_s')).to(dev)
    net.train()
    rs = np.random.RandomState(11)
    N, B, K, A = 7, 3, 5, 6
    tr = Transition(rs.uniform(0, 1, (N, 4, 96, 96)).astype(np.float32), rs.randint(0, A, (N, K)).astype(np.int8), rs.dirichlet(np.ones(A), size=(N, K)).astype(np.float32),
                    rs.uniform(-5, 5, (N, K)).astype(np.float32), rs.uniform(-1, 1, (N, K)).astype(np.float32))
    idx = np.array([5, 0, 3])
    w = torch.from_numpy(rs.uniform(0.3, 1, B).astype(np.float32)).to(dev)
    hl = _hip(net, dev, 4)
    l1, p1 = hl.grad(_ring(tr, dev), torch.from_numpy(idx).to(dev), w, B)
    g1, l1, p1 = hl.grad_flat.clone(), l1.clone(), p1.clone()
    l2, p2 = hl.grad(_ring(Transition(*[x[idx] for x in tr]), dev), None, w, B)
    assert torch.equal(g1, hl.grad_flat) and torch.equal(l1, l2) and torch.equal(p1, p2)
    l3, p3 = hl.grad(_ring(tr, dev), torch.from_numpy(idx).to(dev), w, B)
    assert torch.equal(g1, hl.grad_flat) and torch.equal(l1, l3)


def test_inner_only_tile_convolutions_are_bit_identical_to_whole_haloed_tiles():
    """Round 6: the tiled stages' stride-1 convolutions stage the tile WITH its halo and compute the inner 12 x 12 / 12 x 16 outputs only (`halo_in`,
    a quarter fewer MFMAs); every output is the same chain in the same order as in round 5's whole-tile form (`MZLC_NO_HALO_IN=1`), so loss,
    priorities and the whole gradient vector are the same BITS."""
    import os

    dev = torch.device('cuda', 0)
    rs = np.random.RandomState(21)
    B, K, A = 3, 5, 4  # (conv_case('atari_m'): 8 x 96 x 96 frames, 4 actions, 16 planes, 2 blocks)
    tr = Transition(rs.uniform(0, 1, (B, 8, 96, 96)).astype(np.float32), rs.randint(0, A, (B, K)).astype(np.int8), rs.dirichlet(np.ones(A), size=(B, K)).astype(np.float32),
                    rs.uniform(-5, 5, (B, K)).astype(np.float32), rs.uniform(-1, 1, (B, K)).astype(np.float32))
    w = torch.from_numpy(rs.uniform(0.3, 1, B).astype(np.float32)).to(dev)
    out = []
    # (every form with MZLC_NO_OUT_PLANE: the plane-writing epilogue sums the BatchNorm statistics per TILE, the scatter per 32 positions --
    # the same numbers in another order; it is held to rounding against these below.  NO_KEEP_TILES: the weight gradient's x tiles gathered
    # again instead of kept from the forward pass)
    for flags in (('MZLC_NO_OUT_PLANE',), ('MZLC_NO_OUT_PLANE', 'MZLC_NO_HALO_IN'), ('MZLC_NO_OUT_PLANE', 'MZLC_NO_KEEP_TILES'), (), ('MZLC_KEEP_H1',)):
        for f in flags:
            os.environ[f] = '1'
        try:
            net = build_conv(conv_case('atari_m')).to(dev)
            net.train()
            hl = _hip(net, dev, 4)
            loss, prio = hl.grad(_ring(tr, dev), None, w, B)
            out.append((loss.clone(), prio.clone(), hl.grad_flat.clone()))
            hl.close()
        finally:
            for f in flags:
                os.environ.pop(f, None)
    for o in out[1:3]:
        assert torch.equal(out[0][0], o[0]) and torch.equal(out[0][1], o[1]) and torch.equal(out[0][2], o[2])
    # the default build (outputs written straight into the plane by the conv's epilogue, statistics per tile): same arithmetic but for the order
    # of the BatchNorm partial sums
    # (MZLC_KEEP_H1: the blocks' inner activation as a plane too, masks read from it -- the default forms it in the gather and masks by the sign of
    # a y + b: the same values, the same bits)
    assert torch.equal(out[3][0], out[4][0]) and torch.equal(out[3][1], out[4][1]) and torch.equal(out[3][2], out[4][2])
    l0, p0, g0 = out[0]
    l1, p1, g1 = out[3]
    assert abs(float(l1) - float(l0)) <= 2e-6 * abs(float(l0))
    assert float((p1 - p0).abs().max()) <= 1e-5 * float(p0.abs().max())
    assert float((g1 - g0).abs().max()) <= 2e-4 * float(g0.abs().max())


def test_checkpoint_round_trip_and_planner_epoch():
    """state_dict() / optimizer_state_dict() of a stepped Atari learner restore into a fresh one that then takes bit-identical steps; every
    commit bumps the weights epoch the planner's cache keys on."""
    dev = torch.device('cuda', 0)
    net = build_conv(conv_case('atari_s')).to(dev)
    net.train()
    rs = np.random.RandomState(3)
    B, K, A = 2, 5, 6
    tr = Transition(rs.uniform(0, 1, (B, 4, 96, 96)).astype(np.float32), rs.randint(0, A, (B, K)).astype(np.int8), rs.dirichlet(np.ones(A), size=(B, K)).astype(np.float32),
                    rs.uniform(-5, 5, (B, K)).astype(np.float32), rs.uniform(-1, 1, (B, K)).astype(np.float32))
    hl = _hip(net, dev, B, lr=2e-3, weight_decay=1e-4)
    e0 = getattr(net, '_mz_weights_epoch', 0)
    for _ in range(2):
        hl.step_transitions(tr)
    assert getattr(net, '_mz_weights_epoch', 0) > e0
    sd, osd = copy.deepcopy(net.state_dict()), copy.deepcopy(hl.optimizer_state_dict())
    net2 = build_conv(conv_case('atari_s')).to(dev)
    net2.train()
    hl2 = _hip(net2, dev, B, lr=2e-3, weight_decay=1e-4)
    hl2.load_state_dict(sd)
    hl2.load_optimizer_state_dict(osd)
    hl.step_transitions(tr)
    hl2.step_transitions(tr)
    for k, v in net.state_dict().items():
        assert torch.equal(v, net2.state_dict()[k]), k


def _atari_closed_loop(seed, iters=3, envs=16):
    """device self-play on the synthetic Atari stand-in (C4's kernels on a small net) -> device epilogue -> HBM replay -> HipLearner updates on
    the Atari net -> planner reload, in event order."""
    from muzero_amd import learner as L
    from muzero_amd import planner as pl
    from muzero_amd.config import make_atari_config
    from muzero_amd.network import MuZeroAtariNet
    from muzero_amd.replay import PrioritizedReplay

    dev = torch.device('cuda', 0)
    torch.manual_seed(seed)
    cfg = make_atari_config(batch_size=8, min_replay_size=8, use_tensorboard=False)
    cfg.num_simulations, cfg.num_planes, cfg.num_res_blocks, cfg.value_support_size, cfg.reward_support_size = 6, 8, 1, 11, 11
    cfg.acc_seq_length, cfg.lr_init = 12, 1e-3
    net = MuZeroAtariNet((4, 96, 96), 6, cfg.num_res_blocks, cfg.num_planes, cfg.value_support_size, cfg.reward_support_size).to(dev)
    hl = L.make_hip_learner(cfg, net, dev)
    replay = PrioritizedReplay(512, 0.0, 0.0, np.random.RandomState(seed), device='cuda')
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), cfg, num_envs=envs, seed=seed), 0)
    p.load_state_dict(net.state_dict())
    p.attach_replay(replay, cfg, obs_shape=(4, 96, 96))
    p.selfplay_reset(pl.ENV_SYNTHETIC)
    losses = []
    for _ in range(iters):
        p.selfplay_step(1.0, 14)
        p.synchronize()
        if replay.size < cfg.min_replay_size:
            continue
        for _ in range(2):
            idx, _, ring = replay.sample_indices(cfg.batch_size)
            loss, _ = hl.step(ring, torch.from_numpy(idx).to(dev), None, cfg.batch_size)
            losses.append(loss.clone())
        p.load_state_dict(net.state_dict())
    # the planner now answers with the LEARNER's weights: a fresh planner loaded from the module agrees bit for bit
    obs = np.random.RandomState(1).uniform(0, 1, (3, 4, 96, 96)).astype(np.float32)
    h1, pi1, v1 = p.initial_inference(obs)
    q = pl.Planner(pl.make_mz_config(net.planner_spec(), cfg, num_envs=envs, seed=seed), 0)
    q.load_state_dict(net.state_dict())
    h2, pi2, v2 = q.initial_inference(obs)
    same = np.array_equal(np.asarray(h1), np.asarray(h2)) and np.array_equal(pi1, pi2) and np.array_equal(v1, v2)
    out = ([float(x) for x in losses], {k: v.cpu().numpy().copy() for k, v in net.state_dict().items()}, replay.size, same)
    p.close()
    q.close()
    hl.close()
    return out


def test_atari_closed_loop_runs_and_repeats():
    """Self-play -> replay -> Atari learner -> planner reload on one GPU: finite losses, weights that move, a planner that serves the learner's
    weights, and the same run from the same seed (no atomics anywhere on the path)."""
    a, b = _atari_closed_loop(3), _atari_closed_loop(3)
    assert len(a[0]) >= 4 and np.isfinite(a[0]).all() and a[3]
    assert a[0] == b[0] and a[2] == b[2]
    for k in a[1]:
        assert np.array_equal(a[1][k], b[1][k]), k


def test_full_size_atari_config_at_batch_128_kink_free():
    """make_atari_config's network (128 planes, 8 blocks, supports 61, unroll 5) at ITS batch size -- the configuration bench.py times, where the
    launch-time-aware tilings and chunk counts differ from the small cases above -- against float64 autograd with kink-free weights."""
    dev = torch.device('cuda', 0)
    errs, probe = _case(4, 128, 8, 6, 61, 61, 128, 5, 11, True, dev)
    assert probe.closest_all > 1e-4
    k, e, bar = kinkfree_worst(errs, probe.err32, probe.closest_tie)
    assert e <= bar, (k, e, bar, probe.closest_tie)
