"""SURVEY 8 f2, second stage: the learner step as hand-written HIP kernels (muzero_amd/csrc/mz_learn.h behind include/mzlearner.h)
against (1) the vectors recorded from the REFERENCE's calc_loss / backward / clip / Adam / MultiStepLR (tests/golden/learn_cases.npz,
generator oracle/gen_golden.py learn) and (2) this repo's PyTorch-autograd learner (muzero_amd.learner, itself pinned to the same
vectors by tests/test_learner.py) on the benchmark shapes.  Tolerances are the ones tests/test_gpu_learner.py uses for the autograd
step on the GPU: loss 1e-4 relative, priorities 1e-3, gradients 2e-3 relative (fp32, different summation orders)."""
import copy

import numpy as np
import pytest
import torch

from helpers import build_mlp, load_golden, mlp_case
from muzero_amd import learner
from muzero_amd.replay import Transition

pytestmark = pytest.mark.gpu
G = load_golden('learn_cases.npz')


def _hip(net, dev, max_batch, **kw):
    from muzero_amd.hip_learner import HipLearner

    kw.setdefault('lr', 1e-3)
    return HipLearner(net, dev, 5, max_batch, **kw)


@pytest.mark.parametrize('name,cname', [('mlp_cat', 'tiny'), ('mlp_mse', 'tiny_mse')])
def test_loss_gradients_and_three_updates_match_the_reference(name, cname):
    """The recipe of gen_golden.gen_learn: Adam(lr 1e-3), MultiStepLR([2], 0.1), clip_grad_norm_(10) on the second step only."""
    pre = f'learn_{name}'
    dev = torch.device('cuda', 0)
    net = build_mlp(mlp_case(cname)).to(dev)
    net.train()
    hl = _hip(net, dev, 16, lr=1e-3, milestones=[2], gamma=0.1, max_grad_norm=10.0)
    tr = Transition(*[G[f'{pre}_{f}'] for f in Transition._fields])
    B = tr.state.shape[0]
    ring = dict(state=torch.from_numpy(tr.state).to(dev).reshape(B, -1).contiguous(), action=torch.from_numpy(tr.action).to(dev),
                pi_prob=torch.from_numpy(tr.pi_prob).to(dev), value=torch.from_numpy(tr.value).to(dev), reward=torch.from_numpy(tr.reward).to(dev))
    w = torch.from_numpy(G[f'{pre}_weights']).to(dev)
    losses = []
    for step in range(3):
        loss, prio = hl.grad(ring, None, w, B)
        if step == 0:
            np.testing.assert_allclose(prio.cpu().numpy(), G[f'{pre}_prio'], rtol=1e-3, atol=1e-3)
            for pn in hl.views:
                np.testing.assert_allclose(hl.grad_views[pn].cpu().numpy(), G[f'{pre}_grad_{pn}'], rtol=2e-3, atol=2e-6, err_msg=pn)
        hl.apply(clip=(step == 1))
        losses.append(float(loss))
    np.testing.assert_allclose(losses, G[f'{pre}_losses'], rtol=1e-4)
    sd = net.state_dict()  # the module's parameters ARE the learner's weights
    for pn in hl.views:
        np.testing.assert_allclose(sd[pn].cpu().numpy(), G[f'{pre}_final_{pn}'], rtol=2e-3, atol=2e-5, err_msg=pn)
    assert abs(hl.current_lr() - 1e-4) < 1e-12 and hl.steps == 3


def _random_batch(rs, B, obs_shape, A, K=5, int8_state=False, vmax=50.0):
    st = rs.randint(0, 2, (B,) + obs_shape).astype(np.int8) if int8_state else rs.uniform(-1, 1, (B,) + obs_shape).astype(np.float32)
    return Transition(st, rs.randint(0, A, (B, K)).astype(np.int8), rs.dirichlet(np.ones(A), size=(B, K)).astype(np.float32),
                      rs.uniform(-vmax, vmax, (B, K)).astype(np.float32), rs.uniform(-1, 1, (B, K)).astype(np.float32))


# (lunar with 37 samples has a representation-layer pre-activation of 1.6e-7 -- sample 33, unit 221: its ReLU gate depends on the
# summation order, and the whole row's gradient with it; the test stops at 33 samples)
SHAPES = [('cartpole', 128, False, 50.0), ('cartpole', 100, False, 50.0), ('lunar', 33, False, 200.0), ('tictactoe', 128, True, 1.0), ('odd', 21, False, 5.0)]


@pytest.mark.parametrize('cname,B,int8_state,vmax', SHAPES, ids=[f'{s[0]}-{s[1]}' for s in SHAPES])
def test_step_matches_the_autograd_learner(cname, B, int8_state, vmax):
    """Same start, same batches: six updates (clipping on, weight decay on, an LR milestone inside) of the HIP learner and of
    learner.train_step (PyTorch-ROCm autograd + torch.optim.Adam)."""
    case = mlp_case(cname)
    dev = torch.device('cuda', 0)
    net_a = build_mlp(case).to(dev)
    net_b = copy.deepcopy(net_a)
    net_a.train()
    net_b.train()
    cfg = type('Cfg', (), dict(clip_grad=True, max_grad_norm=5.0))()
    opt = torch.optim.Adam(net_a.parameters(), lr=2e-3, weight_decay=1e-4)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[3], gamma=0.1)
    hl = _hip(net_b, dev, B, lr=2e-3, weight_decay=1e-4, milestones=[3], gamma=0.1, clip_grad=True, max_grad_norm=5.0)
    rs = np.random.RandomState(7)
    for step in range(6):
        tr = _random_batch(rs, B, tuple(case[1]), case[2], int8_state=int8_state, vmax=vmax)
        w = rs.uniform(0.3, 1.0, B).astype(np.float32)
        tr_a = tr._replace(state=tr.state.astype(np.float32))
        if step == 0:  # gradients of the first batch, before anything moves
            opt.zero_grad()
            la0, _ = learner.calc_loss(net_a, dev, tr_a, torch.from_numpy(w).to(dev))
            la0.backward()
            ga = {k: p.grad.detach().clone() for k, p in net_a.named_parameters()}
            B_ = tr.state.shape[0]
            ring = dict(state=torch.from_numpy(tr.state).to(dev).reshape(B_, -1).contiguous(), action=torch.from_numpy(tr.action).to(dev),
                        pi_prob=torch.from_numpy(tr.pi_prob).to(dev), value=torch.from_numpy(tr.value).to(dev), reward=torch.from_numpy(tr.reward).to(dev))
            lb0, _ = hl.grad(ring, None, torch.from_numpy(w).to(dev), B_)
            assert abs(float(la0) - float(lb0)) <= 1e-4 * max(1.0, abs(float(la0)))
            for k in ga:
                a, b = ga[k].cpu().numpy(), hl.grad_views[k].cpu().numpy()
                scale = max(1e-8, float(np.abs(a).max()))
                assert float(np.abs(a - b).max()) <= 2e-3 * scale, (k, float(np.abs(a - b).max()), scale)
        la, pa = learner.train_step(cfg, net_a, opt, sch, dev, tr_a, w)
        lb, pb = hl.step_transitions(tr, w)
        assert abs(la - float(lb)) <= 2e-4 * max(1.0, abs(la)), (step, la, float(lb))
        np.testing.assert_allclose(pb.cpu().numpy(), pa, rtol=2e-3, atol=2e-3 * max(1.0, vmax / 10))
        assert abs(sch.get_last_lr()[0] - hl.current_lr()) < 1e-12
    for (n, x), (_, y) in zip(net_a.state_dict().items(), net_b.state_dict().items()):
        assert float((x - y).abs().max()) < 1e-3, n  # weights move by ~lr per update: six updates of 2e-3


def test_batch_read_from_the_replay_ring_by_index_equals_the_stacked_batch():
    """The kernels gather sampled items from the HBM ring themselves (rows by index, repeated rows included); identical bits to
    the same items stacked contiguously, and the gradient slices of a split reduction (k_learn_dw_big) add up to the unsplit gradient."""
    case = mlp_case('cartpole')
    dev = torch.device('cuda', 0)
    rs = np.random.RandomState(3)
    cap, B = 300, 208  # 13 tiles x 5 steps = 65 reduction blocks: with grad_slices > 1 the 4 x 4-tile kernel of long reductions runs
    items = _random_batch(rs, cap, (4, 5), 2)
    ring = {f: torch.from_numpy(np.ascontiguousarray(getattr(items, f))).to(dev) for f in Transition._fields}
    ring['state'] = ring['state'].reshape(cap, -1).contiguous()
    idx = torch.from_numpy(rs.randint(0, cap, B).astype(np.int64)).to(dev)
    idx[5] = idx[4]
    w = torch.from_numpy(rs.uniform(0.5, 1, B).astype(np.float32)).to(dev)
    res = []
    for mode in ('ring', 'stacked', 'ring', 'sliced'):
        net = build_mlp(case).to(dev)
        hl = _hip(net, dev, B, grad_slices=3 if mode == 'sliced' else 1)
        if mode == 'stacked':
            sub = {f: ring[f].index_select(0, idx).contiguous() for f in ring}
            loss, prio = hl.grad(sub, None, w, B)
        else:
            loss, prio = hl.grad(ring, idx, w, B)
        res.append((float(loss), prio.cpu().numpy().copy(), hl.grad_flat.cpu().numpy().copy()))
        hl.close()
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert res[0][0] == res[2][0] and np.array_equal(res[0][2], res[2][2])  # run to run: bit-identical (no atomics anywhere)
    assert res[0][0] == res[3][0]
    np.testing.assert_allclose(res[3][2], res[0][2], rtol=1e-4, atol=1e-7)


def test_errors_are_reported_not_swallowed():
    from muzero_amd.hip_learner import LearnerError

    dev = torch.device('cuda', 0)
    net = build_mlp(mlp_case('tiny')).to(dev)
    hl = _hip(net, dev, 8)
    tr = _random_batch(np.random.RandomState(0), 12, (3, 4), 3)
    with pytest.raises(LearnerError, match='max_batch'):
        hl.step_transitions(tr)
    bad = tr._replace(pi_prob=tr.pi_prob[:, :, :2])
    with pytest.raises(LearnerError):
        hl.step_transitions(Transition(*[x[:8] for x in bad]))
    # a host-resident replay ring (PrioritizedReplay's default device) must be refused, not read as device addresses
    ok = Transition(*[x[:8] for x in tr])
    host_ring = {f: torch.from_numpy(np.ascontiguousarray(getattr(ok, f))) for f in Transition._fields}
    host_ring['state'] = host_ring['state'].reshape(8, -1).contiguous()
    with pytest.raises(LearnerError, match="device='cuda'"):
        hl.grad(host_ring, None, None, 8)
    # ... and so does the C ABI itself, for callers that do not go through hip_learner.py: a host pointer in mzl_batch
    import ctypes as C

    from muzero_amd.hip_learner import MzlBatch, load_library

    dring = {f: t.to(dev) for f, t in host_ring.items()}
    idx, w = hl._iota[:8], hl._ones
    b = MzlBatch(host_ring['state'].data_ptr(), dring['action'].data_ptr(), dring['pi_prob'].data_ptr(), dring['value'].data_ptr(), dring['reward'].data_ptr(),
                 idx.data_ptr(), w.data_ptr(), hl.loss.data_ptr(), hl.priorities.data_ptr(), 8, 0, 1)
    L = load_library()
    assert L.mzl_grad(hl._h, C.byref(b), None) == -1 and b'd_state is not memory' in L.mzl_last_error()
    # the Atari kernels cut the 48 x 48 and 24 x 24 stages into 12 x 12 tiles: frames other than the reference's 96 x 96 are refused, not mis-tiled
    from muzero_amd.network import MuZeroAtariNet

    with pytest.raises(LearnerError, match='96'):
        _hip(MuZeroAtariNet((4, 64, 64), 6, 1, 8, 11, 11).to(dev), dev, 8)


def _closed_loop(seed, iters=6, envs=64):
    """device self-play -> device epilogue -> HBM replay -> HipLearner updates (batch gathered by index) -> planner reload, in event
    order (the planner's stream is drained before a batch is drawn)."""
    from muzero_amd import learner as L
    from muzero_amd import planner as pl
    from muzero_amd.config import make_classic_config
    from muzero_amd.network import MuZeroMLPNet
    from muzero_amd.replay import PrioritizedReplay

    dev = torch.device('cuda', 0)
    torch.manual_seed(seed)
    cfg = make_classic_config(batch_size=64, min_replay_size=64, use_tensorboard=False)
    cfg.num_simulations = 10
    net = MuZeroMLPNet((4, 5), 2, cfg.num_planes, cfg.value_support_size, cfg.reward_support_size, cfg.hidden_dim).to(dev)
    hl = L.make_hip_learner(cfg, net, dev)
    replay = PrioritizedReplay(20000, 0.0, 0.0, np.random.RandomState(seed), device='cuda')
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), cfg, num_envs=envs, seed=seed), 0)
    p.load_state_dict(net.state_dict())
    p.attach_replay(replay, cfg, obs_shape=(4, 5))
    p.selfplay_reset(pl.ENV_CARTPOLE)
    losses = []
    for _ in range(iters):
        p.selfplay_step(1.0, 24)
        p.synchronize()
        if replay.size < cfg.min_replay_size:
            continue
        for _ in range(4):
            idx, _, ring = replay.sample_indices(cfg.batch_size)
            loss, _ = hl.step(ring, torch.from_numpy(idx).to(dev), None, cfg.batch_size)
            losses.append(loss.clone())
        p.load_state_dict(net.state_dict())
    out = ([float(x) for x in losses], {k: v.cpu().numpy().copy() for k, v in net.state_dict().items()}, replay.size)
    p.close()
    hl.close()
    return out


def test_closed_loop_is_reproducible_from_its_seed():
    """the same seed must give the same run.  Philox self-play keyed by (seed, env, move), a replay draw after the
    committed count is final, and a learner without atomics: losses and final weights are bit-identical; another seed differs."""
    a, b, c = _closed_loop(5), _closed_loop(5), _closed_loop(6)
    assert len(a[0]) >= 8 and np.isfinite(a[0]).all()
    assert a[0] == b[0] and a[2] == b[2]
    for k in a[1]:
        assert np.array_equal(a[1][k], b[1][k]), k
    assert a[0] != c[0]


def test_run_training_drives_the_hip_learner(tmp_path):
    """learner.run_training (pipeline.py:170-286) with make_hip_learner's optimizer / scheduler views: checkpoints in the reference's
    dict layout (optimizer state in torch.optim.Adam's format), actor refresh, stop protocol."""
    import queue
    import threading
    import types

    from muzero_amd.config import make_tictactoe_config
    from muzero_amd.pipeline import load_checkpoint
    from muzero_amd.replay import PrioritizedReplay

    dev = torch.device('cuda', 0)
    net, actor = build_mlp(mlp_case('tictactoe')).to(dev), build_mlp(mlp_case('tictactoe'))
    cfg = make_tictactoe_config(num_training_steps=6, batch_size=32, min_replay_size=48, use_tensorboard=False)
    cfg.checkpoint_interval = 3
    hl = learner.make_hip_learner(cfg, net, dev)
    rs = np.random.RandomState(0)
    rp = PrioritizedReplay(256, 0.0, 0.0, np.random.RandomState(1), device='cuda')
    items = _random_batch(rs, 64, (9, 3, 3), 10, vmax=1.0)
    rp.add_batch(items._replace(state=items.state.reshape(64, 9, 3, 3)), np.ones(64))
    counter, stop, files = types.SimpleNamespace(value=0), threading.Event(), []
    before = {k: v.clone() for k, v in net.state_dict().items()}
    learner.run_training(cfg, net, hl.optimizer, hl.lr_scheduler, dev, actor, rp, queue.SimpleQueue(), counter, str(tmp_path), files, stop, stop_grace_seconds=0.0)
    assert stop.is_set() and counter.value == 6 and len(files) == 2 and hl.steps == 6
    assert any(not torch.equal(before[k], v) for k, v in net.state_dict().items())
    for k, v in actor.state_dict().items():
        assert torch.equal(v.cpu(), net.state_dict()[k].cpu())
    ck = load_checkpoint(str(tmp_path / 'train_steps_6_final'), torch.device('cpu'))
    assert set(ck) == {'network', 'optimizer', 'lr_scheduler', 'train_steps'} and ck['train_steps'] == 6
    # the file holds the weights and Adam's two moments once each -- not the learner's flat vector (weights + moments + gradient slices)
    # behind every parameter view
    import os
    nbytes = 4 * sum(v.numel() for v in net.state_dict().values())
    assert os.path.getsize(tmp_path / 'train_steps_6_final') < 3 * nbytes + (1 << 16)
    # the optimizer entry loads into a torch Adam built over the same parameters (the reference's resume path, pipeline.py:810-817)
    ref = build_mlp(mlp_case('tictactoe'))
    opt = torch.optim.Adam(ref.parameters(), lr=cfg.lr_init, weight_decay=cfg.weight_decay)
    opt.load_state_dict(ck['optimizer'])
    assert float(opt.state[next(iter(ref.parameters()))]['step']) == 6.0


def test_plane_sliced_chains_equal_the_unsliced_stages(monkeypatch):
    """Small batches run the forward / backward chains cut four ways across the planes (k_learn_fwd_sliced / k_learn_back_sliced: partial
    second-layer sums added by the consumer stage); MZL_NO_SLICE=1 keeps one workgroup per tile.  Same loss and gradients up to the
    summation order of the partials."""
    case = mlp_case('cartpole')
    dev = torch.device('cuda', 0)
    rs = np.random.RandomState(11)
    B = 112
    tr = _random_batch(rs, B, (4, 5), 2)
    ring = {f: torch.from_numpy(np.ascontiguousarray(getattr(tr, f))).to(dev) for f in Transition._fields}
    ring['state'] = ring['state'].reshape(B, -1).contiguous()
    w = torch.from_numpy(rs.uniform(0.5, 1, B).astype(np.float32)).to(dev)
    out = []
    for no_slice in ('0', '1'):
        if no_slice == '1':
            monkeypatch.setenv('MZL_NO_SLICE', '1')
        hl = _hip(build_mlp(case).to(dev), dev, B)
        loss, prio = hl.grad(ring, None, w, B)
        out.append((float(loss), prio.cpu().numpy().copy(), hl.grad_flat.cpu().numpy().copy()))
        hl.close()
    assert abs(out[0][0] - out[1][0]) <= 1e-6 * abs(out[1][0])
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-5, atol=1e-5)
    scale = np.abs(out[1][2]).max()
    assert np.abs(out[0][2] - out[1][2]).max() <= 1e-5 * scale
    assert not np.array_equal(out[0][2], out[1][2])  # (the two paths really are different kernels)


def test_persistent_chain_kernels_equal_the_stage_kernels(monkeypatch):
    """From 96 tiles on the dynamics chain runs in two persistent kernels (k_learn_dyn_chain / k_learn_dyn_back_chain: one workgroup per
    CU keeps the dynamics net's operands in registers across the K stages of its tiles, h_k passing through LDS); below that, and with
    MZL_CHAIN_MIN_TILES out of reach, one stage kernel per step.  Same arithmetic in the same order: bit-identical loss, priorities and
    gradients -- also when a workgroup loops over several tiles (more tiles than CUs)."""
    case = mlp_case('cartpole')
    dev = torch.device('cuda', 0)
    rs = np.random.RandomState(12)
    for B in (1600, 4800):  # 100 tiles (one per workgroup) and 300 (256 CUs: some workgroups run two tiles)
        tr = _random_batch(rs, B, (4, 5), 2)
        ring = {f: torch.from_numpy(np.ascontiguousarray(getattr(tr, f))).to(dev) for f in Transition._fields}
        ring['state'] = ring['state'].reshape(B, -1).contiguous()
        w = torch.from_numpy(rs.uniform(0.5, 1, B).astype(np.float32)).to(dev)
        out = []
        for chain in (True, False):
            monkeypatch.setenv('MZL_NO_SLICE', '1')  # (the plane-sliced stages add their partials in another order)
            monkeypatch.setenv('MZL_CHAIN_MIN_TILES', '96' if chain else '1000000')
            monkeypatch.setenv('MZL_CHAIN_FAST_MAX_TILES', '1000000')  # the register-resident stage kernels at every size
            hl = _hip(build_mlp(case).to(dev), dev, B, grad_slices=1)
            loss, prio = hl.grad(ring, None, w, B)
            out.append((float(loss), prio.cpu().numpy().copy(), hl.grad_flat.cpu().numpy().copy()))
            hl.close()
        assert out[0][0] == out[1][0]
        np.testing.assert_array_equal(out[0][1], out[1][1])
        np.testing.assert_array_equal(out[0][2], out[1][2])


def test_resume_from_a_checkpoint_continues_the_same_run():
    """The reference's resume path (pipeline.py:810-817): `network.load_state_dict`, `optimizer.load_state_dict`, `lr_scheduler.load_state_dict`
    on freshly built objects, then training goes on.  Six updates in one go == three updates, a checkpoint through torch.save's format, three more
    on a NEW learner -- bit for bit (the loaded weights reach the kernels' operand copies although torch wrote them, Adam's moments and step
    count and the schedule's position travel in torch's own formats)."""
    import io

    case = mlp_case('cartpole')
    dev = torch.device('cuda', 0)
    rs = np.random.RandomState(5)
    B = 48
    batches = []
    for _ in range(6):
        tr = _random_batch(rs, B, (4, 5), 2)
        ring = {f: torch.from_numpy(np.ascontiguousarray(getattr(tr, f))).to(dev) for f in Transition._fields}
        ring['state'] = ring['state'].reshape(B, -1).contiguous()
        batches.append(ring)
    kw = dict(lr=1e-2, weight_decay=1e-4, milestones=(2, 4), gamma=0.5, clip_grad=True, max_grad_norm=1.0)
    net_a = build_mlp(case).to(dev)
    hl_a = _hip(net_a, dev, B, **kw)
    for ring in batches:
        hl_a.step(ring, None, None, B, allreduce=False)
    net_b = build_mlp(case).to(dev)
    hl_b = _hip(net_b, dev, B, **kw)
    for ring in batches[:3]:
        hl_b.step(ring, None, None, B, allreduce=False)
    buf = io.BytesIO()
    torch.save({'network': net_b.state_dict(), 'optimizer': hl_b.optimizer.state_dict(), 'lr_scheduler': hl_b.lr_scheduler.state_dict(), 'train_steps': 3}, buf)
    hl_b.close()
    buf.seek(0)
    ck = torch.load(buf, map_location='cpu', weights_only=False)
    torch.manual_seed(123)
    from muzero_amd.network import MuZeroMLPNet
    net_c = MuZeroMLPNet((4, 5), 2, case[3], case[4], case[5], case[6]).to(dev)  # fresh random weights, replaced by the checkpoint's
    hl_c = _hip(net_c, dev, B, **kw)
    net_c.load_state_dict(ck['network'])
    hl_c.optimizer.load_state_dict(ck['optimizer'])
    hl_c.lr_scheduler.load_state_dict(ck['lr_scheduler'])
    assert hl_c.steps == 3 and hl_c.current_lr() == hl_a.current_lr(3)
    for ring in batches[3:]:
        hl_c.step(ring, None, None, B, allreduce=False)
    for (k, x), (_, y) in zip(net_a.state_dict().items(), net_c.state_dict().items()):
        assert torch.equal(x, y), k
    oa, oc = hl_a.optimizer.state_dict(), hl_c.optimizer.state_dict()
    for i in oa['state']:
        assert torch.equal(oa['state'][i]['exp_avg'], oc['state'][i]['exp_avg']) and torch.equal(oa['state'][i]['exp_avg_sq'], oc['state'][i]['exp_avg_sq'])


def test_load_state_dict_through_the_learner_reaches_the_inference_engine():
    """HipLearner.load_state_dict / adopt write the master weights through views of the flat vector, which bumps neither the
    parameters' version counters nor -- until round 5 -- the module's weights epoch: initial_inference / uct_search on the adopted module kept
    serving the previous weights.  Now commit() bumps the epoch: inference after load_state_dict equals a fresh net loaded with the same weights."""
    dev = torch.device('cuda', 0)
    net = build_mlp(mlp_case('cartpole')).to(dev)
    other = build_mlp(('other', (4, 5), 2, 512, 31, 31, 64, 99)).to(dev)
    x = torch.rand(1, 4, 5, device=dev)
    before = net.initial_inference(x)  # binds the module's engine to the current weights
    hl = _hip(net, dev, 8)
    assert net.initial_inference(x).value == before.value
    hl.load_state_dict(other.state_dict())
    after, want = net.initial_inference(x), other.initial_inference(x)
    assert after.value != before.value
    assert after.value == want.value and np.array_equal(after.pi_probs, want.pi_probs) and np.array_equal(after.hidden_state, want.hidden_state)
    # adopt() on a module whose engine is already bound: same rule
    third = build_mlp(('third', (4, 5), 2, 512, 31, 31, 64, 123)).to(dev)
    ref3 = third.initial_inference(x)
    fresh = build_mlp(mlp_case('cartpole')).to(dev)
    _ = fresh.initial_inference(x)
    hl2 = _hip(fresh, dev, 8)
    hl2.load_state_dict(third.state_dict())
    assert fresh.initial_inference(x).value == ref3.value
