"""SURVEY 8 f2, the conv half: the learner step of MuZeroBoardGameNet (residual towers, train-mode BatchNorm) as hand-written HIP kernels
(muzero_amd/csrc/mz_learn_conv.h behind include/mzlearner.h, net_kind MZL_NET_BOARD) against
(1) the vectors recorded from the REFERENCE's calc_loss / backward / clip / Adam / MultiStepLR on its own MuZeroBoardGameNet
    (tests/golden/learn_cases.npz `learn_conv_board3`, generator oracle/gen_golden.py learn): loss 1e-4, gradients 2e-3, three updates,
    BatchNorm running statistics included;
(2) PyTorch-ROCm autograd (muzero_amd.learner, pinned to the same vectors by tests/test_learner.py) in float64 on the same batch over the
    geometries that pick different kernel builds (pixel tilings NPT 6 / 9 / 15, images per workgroup, plane counts off the 16-tile, wide
    action-plane convs, int8 states, int16 actions).
Tolerance (round 6): the float64 reference is told what the HIP forward pass DECIDED -- every ReLU mask, every arg-min / arg-max of
normalize_hidden_state, read back through the library's diagnostic hook (tests/forced_masks.py) -- and takes the same branches, so a float32
pass and a float64 pass are compared on ONE piecewise-linear function: every gradient tensor of every geometry, seeded RANDOM weights, within
1e-4 of its largest entry (measured 4e-5 and below; plain float64 autograd, which decides for itself, differs by 1e-2 .. 1.5e-1 wherever a
pre-activation lies within float32 rounding of zero -- the 8e-2 "kinked" bar and the KNOWN_KINKED_SET of rounds 4-5 are gone).  At full size
(the C5 net, batch 128: 8-block train-mode BatchNorm towers) float32 rounding itself reaches 1.6e-3 in PyTorch-ROCm's own float32 autograd
on the same branch; the kernels are held to 4 x that yardstick there (the factor the kink-free tests use) (measured 2.3e-3).  The kink-free weight construction is kept as an
independent second check (no read-back of the library's tensors involved)."""
import copy
import os

import numpy as np
import pytest
import torch

from helpers import build_conv, conv_case, load_golden, seeded_state_dict
from muzero_amd import learner
from muzero_amd.replay import Transition

pytestmark = pytest.mark.gpu
G = load_golden('learn_cases.npz')


def _hip(net, dev, max_batch, K=5, **kw):
    from muzero_amd.hip_learner import HipLearner

    kw.setdefault('lr', 1e-3)
    return HipLearner(net, dev, K, max_batch, **kw)


def _ring(tr, dev):
    B = tr.state.shape[0]
    return dict(state=torch.from_numpy(tr.state).to(dev).reshape(B, -1).contiguous(), action=torch.from_numpy(tr.action).to(dev),
                pi_prob=torch.from_numpy(tr.pi_prob).to(dev), value=torch.from_numpy(tr.value).to(dev), reward=torch.from_numpy(tr.reward).to(dev))


def test_loss_gradients_and_three_updates_match_the_reference():
    """The recipe of gen_golden.gen_learn on the reference's MuZeroBoardGameNet (3 x 3 board, 16 planes, 2 blocks): Adam(lr 1e-3),
    MultiStepLR([2], 0.1), clip_grad_norm_(10) on the second step only, the network in train mode."""
    pre = 'learn_conv_board3'
    dev = torch.device('cuda', 0)
    net = build_conv(conv_case('board3')).to(dev)
    net.train()
    hl = _hip(net, dev, 16, lr=1e-3, milestones=[2], gamma=0.1, max_grad_norm=10.0)
    tr = Transition(*[G[f'{pre}_{f}'] for f in Transition._fields])
    B = tr.state.shape[0]
    ring = _ring(tr, dev)
    w = torch.from_numpy(G[f'{pre}_weights']).to(dev)
    losses = []
    for step in range(3):
        loss, prio = hl.grad(ring, None, w, B)
        if step == 0:
            np.testing.assert_allclose(prio.cpu().numpy(), G[f'{pre}_prio'], rtol=1e-3, atol=1e-3)
            for pn in hl.views:
                ref = G[f'{pre}_grad_{pn}']
                np.testing.assert_allclose(hl.grad_views[pn].cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * float(np.abs(ref).max()) + 1e-7, err_msg=pn)
        hl.apply(clip=(step == 1))
        losses.append(float(loss))
    np.testing.assert_allclose(losses, G[f'{pre}_losses'], rtol=1e-4)
    sd = net.state_dict()  # the module's parameters AND BatchNorm buffers are the learner's vectors
    for pn in sd:
        ref = G[f'{pre}_final_{pn}']
        np.testing.assert_allclose(sd[pn].cpu().numpy(), ref, rtol=2e-3, atol=2e-5 + 1e-4 * float(np.abs(ref).max()), err_msg=pn)
    assert abs(hl.current_lr() - 1e-4) < 1e-12 and hl.steps == 3


def _net(board, planes, blocks, chan, seed, dev):
    from muzero_amd.network import MuZeroBoardGameNet

    A = board * board + 1
    net = MuZeroBoardGameNet((chan, board, board), A, blocks, planes)
    net.load_state_dict(seeded_state_dict(net, seed))
    return net.to(dev), A


def _batch(rs, B, shape, A, K=5, int8_state=False):
    st = rs.randint(0, 2, (B,) + shape).astype(np.int8) if int8_state else rs.uniform(0, 1, (B,) + shape).astype(np.float32)
    return Transition(st, rs.randint(0, A, (B, K)).astype(np.int16 if A > 128 else np.int8), rs.dirichlet(np.ones(A), size=(B, K)).astype(np.float32),
                      rs.uniform(-1, 1, (B, K)).astype(np.float32), rs.uniform(-1, 1, (B, K)).astype(np.float32))


class _KinkProbe:
    """While active, records how close the float64 forward pass comes to a kink: the smallest |pre-activation| over every ReLU (nn.ReLU calls
    F.relu), and the smallest gap between the two largest / two smallest DISTINCT-POSITION channel values of every normalize_hidden_state
    (a tie at exactly 0 between ReLU zeros is harmless: their gradient is masked either way)."""

    def __init__(self):
        self.closest = float('inf')

    def __enter__(self):
        import torch.nn.functional as F

        from muzero_amd import network as nw

        self._F, self._nw, self._relu, self._norm = F, nw, F.relu, nw.normalize_hidden_state
        probe = self

        def relu(x, inplace=False):
            nz = x.detach().abs()
            probe.closest = min(probe.closest, float(nz[nz > 0].min()) if bool((nz > 0).any()) else float('inf'))
            return probe._relu(x, inplace=False)

        def normalize(h):
            v = h.detach().flatten(2) if h.dim() > 2 else h.detach()
            top = v.topk(2, dim=1).values
            low = (-v).topk(2, dim=1).values
            probe.closest = min(probe.closest, float((top[:, 0] - top[:, 1]).min()))
            gap = (low[:, 0] - low[:, 1]).abs()
            live = (-low[:, 0]) > 0  # the minimum is not a ReLU zero
            if bool(live.any()):
                probe.closest = min(probe.closest, float(gap[live].min()))
            return probe._norm(h)

        F.relu, nw.normalize_hidden_state = relu, normalize
        return self

    def __exit__(self, *exc):
        self._F.relu, self._nw.normalize_hidden_state = self._relu, self._norm


def _f64_reference(net, tr, w, dev):
    net_d = copy.deepcopy(net).double()
    net_d.train()
    t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x)).to(dev, dt)  # noqa: E731
    with _KinkProbe() as probe:
        loss, prio = learner.loss_tensors(net_d, t(tr.state, torch.float64), t(tr.action, torch.int64), t(tr.value, torch.float64), t(tr.reward, torch.float64),
                                          t(tr.pi_prob, torch.float64), t(w, torch.float64))
    loss.backward()
    return float(loss.detach()), prio.detach(), {k: p.grad for k, p in net_d.named_parameters()}, net_d.state_dict(), probe.closest


# board, planes, blocks, observation planes, batch, int8 states
GEOMETRIES = [(3, 16, 2, 9, 4, False), (5, 8, 1, 5, 7, False), (9, 8, 1, 9, 6, False), (9, 32, 3, 9, 64, True), (6, 128, 1, 2, 17, True), (7, 40, 1, 4, 33, False),
              (11, 8, 2, 9, 9, False), (13, 24, 1, 3, 5, False), (15, 16, 1, 9, 3, False), (15, 32, 2, 9, 10, True), (4, 16, 1, 3, 5, False), (8, 16, 1, 2, 6, False),
              (10, 8, 1, 2, 4, False), (12, 16, 1, 2, 3, False), (14, 8, 1, 3, 2, False), (15, 8, 1, 2, 2, False), (5, 256, 1, 3, 6, False)]


SAME_BRANCH_TOL = 1e-4  # of each gradient tensor's largest entry, random weights, against float64 on the HIP pass's own branch (measured: 4e-5 and below)


def _same_branch(hl, net, tr, w, B, K, dev, want_grads=False):
    """float64 autograd told what the HIP forward pass decided at every ReLU and every normalisation (tests/forced_masks.py), PyTorch-ROCm's
    float32 on the same branch as the yardstick of what float32 reaches; returns (errors per tensor, torch-float32 errors per tensor, loss,
    priorities, ReLU decisions that the float64 pass would have taken differently)."""
    from forced_masks import forced_f64, hip_decisions, tensor_errors

    masks, norms = hip_decisions(hl, net, B, K)
    loss_d, prio_d, gd, flipped = forced_f64(net, tr._replace(state=tr.state.astype(np.float64)), w, dev, masks, norms)
    _, _, g32, _ = forced_f64(net, tr._replace(state=tr.state.astype(np.float32)), w, dev, masks, norms, dtype=torch.float32)
    out = (tensor_errors(gd, hl.grad_views), tensor_errors(gd, g32), loss_d, prio_d, flipped)
    return out + (gd,) if want_grads else out


def same_branch_bar(errs, err32, flat=SAME_BRANCH_TOL):
    """(tensor, error, bar) of the worst offender: every tensor within `flat` of its largest entry -- or, where deep train-mode BatchNorm towers
    at a large batch amplify float32 rounding beyond that, within 4 x the worst error PyTorch-ROCm's own float32 autograd shows ON THE SAME
    BRANCH (C5 net at batch 128: PyTorch 1.6e-3, the kernels 2.3e-3; small geometries: both below 5e-5)."""
    bar = max(flat, 4.0 * max(err32.values()))
    k = max(errs, key=errs.get)
    return k, errs[k], bar


@pytest.mark.parametrize('board,planes,blocks,chan,B,int8_state', GEOMETRIES, ids=[f'b{g[0]}-p{g[1]}-r{g[2]}-n{g[4]}' for g in GEOMETRIES])
def test_gradient_matches_float64_autograd(board, planes, blocks, chan, B, int8_state):
    """Seeded RANDOM weights (full-strength element-wise ReLU masks), every geometry, no kink allowance (round 6): the float64 reference takes the
    branches the HIP forward pass took, so what is left is rounding -- every gradient tensor within 1e-4 of its largest entry."""
    dev = torch.device('cuda', 0)
    net, A = _net(board, planes, blocks, chan, 100 + board, dev)
    net.train()
    rs = np.random.RandomState(board * 7 + B)
    tr = _batch(rs, B, (chan, board, board), A, int8_state=int8_state)
    w = rs.uniform(0.3, 1.0, B).astype(np.float32)
    # the running statistics and the loss do not depend on the branch to first order: plain float64 autograd
    loss_p, prio_p, _, sd_d, closest = _f64_reference(net, tr._replace(state=tr.state.astype(np.float64)), w, dev)
    hl = _hip(net, dev, B)
    loss, prio = hl.grad(_ring(tr, dev), None, torch.from_numpy(w).to(dev), B)
    assert abs(float(loss) - loss_p) <= 1e-4 * max(1.0, abs(loss_p))
    np.testing.assert_allclose(prio.cpu().numpy(), prio_p.cpu().numpy(), rtol=1e-3, atol=1e-4)
    errs, err32, loss_d, prio_d, flipped = _same_branch(hl, net, tr, w, B, 5, dev)
    assert abs(float(loss) - loss_d) <= 2e-6 * max(1.0, abs(loss_d))
    k, e, bar = same_branch_bar(errs, err32)
    assert e <= bar, (k, e, bar, 'decisions float64 would have taken differently:', flipped, 'closest pre-activation / gap to a tie', closest)
    sd = net.state_dict()  # the train-mode forward pass has updated the running statistics (network.py:283-291), one step per application
    for k, v in sd_d.items():
        if 'running' in k:
            assert float((v - sd[k].double()).abs().max()) <= 1e-5 * max(1.0, float(v.abs().max())), k
        if 'num_batches_tracked' in k:
            assert int(v) == int(sd[k]), k


@pytest.mark.parametrize('K', [1, 2, 3, 8])
def test_other_unroll_lengths(K):
    """Every reference configuration unrolls 5 steps; the kernels take any K <= 32 (the heads' grouping, the accumulate-in-step-order rule and the
    0.5 / (1 / K) gradient scales all depend on it)."""
    dev = torch.device('cuda', 0)
    net, A = _net(5, 16, 1, 3, 300 + K, dev)
    net.train()
    rs = np.random.RandomState(K)
    B = 5
    tr = _batch(rs, B, (3, 5, 5), A, K=K)
    w = rs.uniform(0.3, 1.0, B).astype(np.float32)
    loss_p, _, _, sd_d, closest = _f64_reference(net, tr._replace(state=tr.state.astype(np.float64)), w, dev)
    hl = _hip(net, dev, B, K=K)
    loss, prio = hl.grad(_ring(tr, dev), None, torch.from_numpy(w).to(dev), B)
    assert abs(float(loss) - loss_p) <= 1e-4 * max(1.0, abs(loss_p))
    errs, err32, _, _, flipped = _same_branch(hl, net, tr, w, B, K, dev)
    k, e, bar = same_branch_bar(errs, err32)
    assert e <= bar, (k, e, bar, K, flipped)
    sd = net.state_dict()
    for k, v in sd_d.items():
        if 'num_batches_tracked' in k:
            assert int(v) == int(sd[k]), k


@pytest.mark.parametrize('board,planes,blocks,chan,B,int8_state', GEOMETRIES, ids=[f'b{g[0]}-p{g[1]}-r{g[2]}-n{g[4]}' for g in GEOMETRIES])
def test_gradient_matches_float64_autograd_kink_free(board, planes, blocks, chan, B, int8_state):
    """The same geometries with weights that keep every ReLU channel on or off for the whole batch (tests/test_gpu_atari_learner.kinkfree_state_dict,
    a mix of live and dead channels): every gradient tensor within 2e-3 of float64 autograd, no kink allowance."""
    from muzero_amd.network import MuZeroBoardGameNet
    from test_gpu_atari_learner import grad_errors, kinkfree_state_dict

    dev = torch.device('cuda', 0)
    A = board * board + 1
    net = MuZeroBoardGameNet((chan, board, board), A, blocks, planes)
    net.load_state_dict(kinkfree_state_dict(net, 100 + board))
    net = net.to(dev)
    net.train()
    rs = np.random.RandomState(board * 7 + B)
    tr = _batch(rs, B, (chan, board, board), A, int8_state=int8_state)
    w = rs.uniform(0.3, 1.0, B).astype(np.float32)
    loss_d, prio_d, gd, sd_d, closest = _f64_reference(net, tr._replace(state=tr.state.astype(np.float64)), w, dev)
    hl = _hip(net, dev, B)
    loss, prio = hl.grad(_ring(tr, dev), None, torch.from_numpy(w).to(dev), B)
    assert abs(float(loss) - loss_d) <= 1e-4 * max(1.0, abs(loss_d))
    np.testing.assert_allclose(prio.cpu().numpy(), prio_d.cpu().numpy(), rtol=1e-3, atol=1e-4)
    errs = grad_errors(gd, hl.grad_views)
    worst = max(errs, key=errs.get)
    assert errs[worst] <= 2e-3, (worst, errs[worst], closest)


def test_six_updates_follow_the_autograd_learner():
    """Same start, same batches: six updates (clipping, weight decay, an LR milestone inside) of the HIP learner and of learner.train_step
    (PyTorch-ROCm autograd + torch.optim.Adam, float64) on a small net -- few enough ReLU elements that the batches stay off the kinks;
    weights and BatchNorm buffers stay together.  (Adam's first steps move every weight by ~lr whatever its gradient's size, so a gradient
    component at rounding distance from 0 goes the other way in a float32 run: a handful of weights may differ by 2 lr per step -- hence a mean and
    a max bar.)"""
    dev = torch.device('cuda', 0)
    net_b, A = _net(5, 8, 1, 5, 77, dev)
    net_a = copy.deepcopy(net_b).double()
    net_a.train()
    net_b.train()
    cfg = type('Cfg', (), dict(clip_grad=True, max_grad_norm=5.0))()
    opt = torch.optim.Adam(net_a.parameters(), lr=2e-3, weight_decay=1e-4)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[3], gamma=0.1)
    B = 12
    hl = _hip(net_b, dev, B, lr=2e-3, weight_decay=1e-4, milestones=[3], gamma=0.1, clip_grad=True, max_grad_norm=5.0)
    rs = np.random.RandomState(5)
    t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x)).to(dev, dt)  # noqa: E731
    for step in range(6):
        tr = _batch(rs, B, (5, 5, 5), A)
        w = rs.uniform(0.3, 1.0, B).astype(np.float32)
        opt.zero_grad()
        la, pa = learner.loss_tensors(net_a, t(tr.state, torch.float64), t(tr.action, torch.int64), t(tr.value, torch.float64), t(tr.reward, torch.float64),
                                      t(tr.pi_prob, torch.float64), t(w, torch.float64))
        la.backward()
        torch.nn.utils.clip_grad_norm_(net_a.parameters(), cfg.max_grad_norm)
        opt.step()
        sch.step()
        lb, pb = hl.step_transitions(tr, w)
        assert abs(float(la) - float(lb)) <= 1e-3 * max(1.0, abs(float(la))), (step, float(la), float(lb))
        np.testing.assert_allclose(pb.cpu().numpy(), pa.detach().cpu().numpy(), rtol=5e-3, atol=5e-3)
        assert abs(sch.get_last_lr()[0] - hl.current_lr()) < 1e-12
    for (n, x), (_, y) in zip(net_a.state_dict().items(), net_b.state_dict().items()):
        d = (x.double() - y.double()).abs()
        sc = max(1.0, float(x.double().abs().max()))
        assert float(d.mean()) < 2e-4 * sc and float(d.max()) < 1.4e-2 * sc, (n, float(d.mean()), float(d.max()))


def test_batch_read_from_the_ring_by_index_and_unpaired_launches_agree_bit_for_bit():
    """(a) rows gathered from the replay ring by an index vector with repeats == the same items stacked; (b) the paired launches (dynamics and
    prediction tower of a step side by side) == one job per launch (MZLC_NO_PAIR=1): no atomics, fixed reduction orders."""
    dev = torch.device('cuda', 0)
    rs = np.random.RandomState(11)
    cap, B = 50, 21
    net, A = _net(15, 16, 1, 5, 31, dev)  # A = 226: int16 actions
    items = _batch(rs, cap, (5, 15, 15), A, int8_state=True)
    ring = _ring(items, dev)
    idx = torch.from_numpy(rs.randint(0, cap, B).astype(np.int64)).to(dev)
    idx[5] = idx[4]
    w = torch.from_numpy(rs.uniform(0.3, 1.0, B).astype(np.float32)).to(dev)
    hl = _hip(copy.deepcopy(net), dev, 32)
    la, pa = hl.grad(ring, idx, w, B)
    ga, la, pa = hl.grad_flat.clone(), la.clone(), pa.clone()
    stacked = {k: v[idx].contiguous() for k, v in ring.items()}
    os.environ['MZLC_NO_PAIR'] = '1'
    try:
        hl2 = _hip(copy.deepcopy(net), dev, B)
    finally:
        del os.environ['MZLC_NO_PAIR']
    lb, pb = hl2.grad(stacked, None, w, B)
    assert torch.equal(la, lb) and torch.equal(pa, pb) and torch.equal(ga, hl2.grad_flat)
    assert torch.equal(hl.running, hl2.running) and torch.equal(hl.num_batches, hl2.num_batches)


def test_weight_gradients_of_all_unroll_steps_in_one_launch_match_the_per_step_launches():
    """Round 6: where one unroll step's batch is a few staging rounds per weight-gradient workgroup (small planes), the block layers of the
    dynamics / prediction towers take their weight gradient from ONE launch per layer over all K steps (LcWgrad::srcs, after the last step's
    backward) instead of K accumulating launches.  Same products, summed in another order: loss and priorities (forward only) are the same
    bits, the gradient agrees to float32 rounding."""
    dev = torch.device('cuda', 0)
    rs = np.random.RandomState(5)
    for board, planes, blocks, chan, B, K in ((6, 64, 2, 3, 32, 5), (9, 32, 3, 4, 17, 3), (3, 16, 1, 2, 8, 5)):
        net, A = _net(board, planes, blocks, chan, 40 + board, dev)
        tr = _batch(rs, B, (chan, board, board), A, K=K)
        w = torch.from_numpy(rs.uniform(0.3, 1.0, B).astype(np.float32)).to(dev)
        out = []
        for flag in ('1', '0'):
            os.environ['MZLC_DEFER_WGRAD'] = flag
            try:
                hl = _hip(copy.deepcopy(net), dev, B, K=K)
            finally:
                del os.environ['MZLC_DEFER_WGRAD']
            la, pa = hl.grad(_ring(tr, dev), None, w, B)
            out.append((la.clone(), pa.clone(), hl.grad_flat.clone()))
            hl.close()
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
        g1, g0 = out[0][2], out[1][2]
        assert float((g1 - g0).abs().max()) <= 2e-6 * float(g0.abs().max()), (board, float((g1 - g0).abs().max()), float(g0.abs().max()))
        assert not torch.equal(g1, g0) or B == 1  # (the two forms really are different launches)


def test_checkpoint_round_trip_and_inference_after_load_state_dict():
    """weights written through the learner (load_state_dict, apply) must reach the module's inference engine; the
    optimizer / scheduler views keep torch's checkpoint format (pipeline.py:224-230)."""
    dev = torch.device('cuda', 0)
    net, A = _net(5, 8, 1, 5, 41, dev)
    other, _ = _net(5, 8, 1, 5, 42, dev)
    net.eval()
    other.eval()
    hl = _hip(net, dev, 8)
    x = torch.rand(1, 5, 5, 5, device=dev)
    before = net.initial_inference(x)
    hl.load_state_dict(other.state_dict())
    after, want = net.initial_inference(x), other.initial_inference(x)
    assert not np.allclose(before.pi_probs, after.pi_probs)
    np.testing.assert_array_equal(after.pi_probs, want.pi_probs)
    assert after.value == want.value
    osd = hl.optimizer.state_dict()
    assert len(osd['state']) == len(list(net.parameters())) and osd['param_groups'][0]['lr'] == 1e-3
    for i, p in enumerate(net.parameters()):
        assert tuple(osd['state'][i]['exp_avg'].shape) == tuple(p.shape)
    sd = net.state_dict()
    assert set(sd) == set(other.state_dict()) and all(torch.equal(sd[k], other.state_dict()[k]) for k in sd)


def test_errors_are_loud():
    from muzero_amd.hip_learner import HipLearner, LearnerError
    from muzero_amd.network import MuZeroBoardGameNet

    dev = torch.device('cuda', 0)
    with pytest.raises(LearnerError):  # 19 x 19 = 361 points: beyond the kernels' whole-image tiling
        HipLearner(MuZeroBoardGameNet((3, 19, 19), 362, 1, 8).to(dev), dev, 5, 4, lr=1e-3)
    net, A = _net(3, 16, 1, 9, 1, dev)
    hl = HipLearner(net, dev, 5, 4, lr=1e-3)
    tr = _batch(np.random.RandomState(0), 4, (9, 3, 3), A)
    with pytest.raises(LearnerError):  # host-resident batch
        hl.grad({k: v.cpu() for k, v in _ring(tr, dev).items()}, None, None, 4)
    with pytest.raises(LearnerError):
        hl.grad(_ring(tr, dev), None, None, 5)  # > max_batch


def test_full_size_c5_net_at_batch_128_random_weights():
    """the C5 network at make_gomoku_config's batch size with seeded RANDOM weights -- the NPT = 15 build, 16 weight-gradient chunks,
    XCD remap and paired launches under full-strength element-wise masks -- against float64 autograd on the HIP pass's own branch
    (tests/forced_masks.py)."""
    dev = torch.device('cuda', 0)
    board, planes, blocks, chan, B = 15, 128, 8, 9, 128
    net, A = _net(board, planes, blocks, chan, 177, dev)
    net.train()
    rs = np.random.RandomState(6)
    tr = _batch(rs, B, (chan, board, board), A, int8_state=True)
    w = rs.uniform(0.3, 1.0, B).astype(np.float32)
    hl = _hip(net, dev, B)
    loss, prio = hl.grad(_ring(tr, dev), None, torch.from_numpy(w).to(dev), B)
    errs, err32, loss_d, prio_d, flipped = _same_branch(hl, net, tr, w, B, 5, dev)
    assert abs(float(loss) - loss_d) <= 3e-5 * max(1.0, abs(loss_d))
    np.testing.assert_allclose(prio.cpu().numpy(), prio_d.cpu().numpy(), rtol=1e-3, atol=1e-4)
    k, e, bar = same_branch_bar(errs, err32)
    print('C5 net, batch 128, random weights: worst tensor', k, '%.2e' % e, 'bar %.2e' % bar, '(PyTorch-ROCm float32 on the same branch: %.2e);' % max(err32.values()),
          flipped, 'ReLU decisions differ from float64\'s own')
    assert e <= bar and bar <= 1e-2, (k, e, bar)
    assert flipped > 0  # (at this size some pre-activation always sits on a kink: the case the plain comparison could not hold tightly)


@pytest.mark.parametrize('board,planes,blocks,chan,B', [(15, 128, 2, 9, 64), (15, 128, 4, 9, 96), (9, 128, 3, 9, 128)], ids=['b15-p128-r2-n64', 'b15-p128-r4-n96', 'b9-p128-r3-n128'])
def test_wide_nets_at_large_batches_random_weights(board, planes, blocks, chan, B):
    """random-weight board-net cases at 128 planes and batches >= 64 (the small geometries stop at 32 planes / batch 64)."""
    dev = torch.device('cuda', 0)
    net, A = _net(board, planes, blocks, chan, 500 + board + blocks, dev)
    net.train()
    rs = np.random.RandomState(board + B)
    tr = _batch(rs, B, (chan, board, board), A)
    w = rs.uniform(0.3, 1.0, B).astype(np.float32)
    hl = _hip(net, dev, B)
    loss, prio = hl.grad(_ring(tr, dev), None, torch.from_numpy(w).to(dev), B)
    errs, err32, loss_d, prio_d, flipped = _same_branch(hl, net, tr, w, B, 5, dev)
    assert abs(float(loss) - loss_d) <= 3e-5 * max(1.0, abs(loss_d))
    k, e, bar = same_branch_bar(errs, err32)
    assert e <= bar and bar <= 5e-3, (k, e, bar, flipped)


def test_full_size_c5_net_at_batch_128_kink_free():
    """The C5 network (15 x 15, 128 planes, 8 blocks, 226 actions, unroll 5) at make_gomoku_config's batch size -- the configuration bench.py times --
    against float64 autograd with kink-free weights: every gradient tensor within the tight bar."""
    from muzero_amd.network import MuZeroBoardGameNet
    from test_gpu_atari_learner import f32_errors, grad_errors, kinkfree_state_dict, kinkfree_worst

    dev = torch.device('cuda', 0)
    board, planes, blocks, chan, B = 15, 128, 8, 9, 128
    A = board * board + 1
    net = MuZeroBoardGameNet((chan, board, board), A, blocks, planes)
    net.load_state_dict(kinkfree_state_dict(net, 77))
    net = net.to(dev)
    net.train()
    rs = np.random.RandomState(5)
    tr = _batch(rs, B, (chan, board, board), A, int8_state=True)
    w = rs.uniform(0.3, 1.0, B).astype(np.float32)
    loss_d, prio_d, gd, sd_d, closest = _f64_reference(net, tr._replace(state=tr.state.astype(np.float64)), w, dev)
    err32 = f32_errors(net, tr._replace(state=tr.state.astype(np.float32)), w, dev, gd)
    hl = _hip(net, dev, B)
    loss, prio = hl.grad(_ring(tr, dev), None, torch.from_numpy(w).to(dev), B)
    assert abs(float(loss) - loss_d) <= 1e-4 * max(1.0, abs(loss_d))
    np.testing.assert_allclose(prio.cpu().numpy(), prio_d.cpu().numpy(), rtol=1e-3, atol=1e-4)
    k, e, bar = kinkfree_worst(grad_errors(gd, hl.grad_views), err32, closest if closest < 1e-6 else float('inf'))
    assert e <= bar, (k, e, bar, closest)
    sd = net.state_dict()
    for kk, v in sd_d.items():
        if 'running' in kk:
            assert float((v - sd[kk].double()).abs().max()) <= 3e-5 * max(1.0, float(v.abs().max())), kk
