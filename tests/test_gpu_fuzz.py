"""Randomised differential test of the MLP search path: configurations drawn over the whole dispatch space -- tuned kernel builds (two /
four / ten actions / general, 256 / 512 planes, categorical / MSE heads, board / single player, with / without known bounds) and the
shape-generic kernel (odd plane counts, hidden sizes, action counts, support sizes, simulation counts) -- every env of every batch bit-exact
against an independent oracle search on injected draws.   MZ_FUZZ_CASES=200 python -m pytest tests/test_gpu_fuzz.py -m gpu   (default 24)"""
import os

import numpy as np
import pytest

from helpers import build_mlp
from test_oracle_nets import _oracle_net


def _planner(net, num_envs, **search):
    from muzero_amd import planner as pl

    p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=num_envs, **search), 0)
    p.load_state_dict(net.state_dict())
    return p


pytestmark = pytest.mark.gpu
CASES = int(os.environ.get('MZ_FUZZ_CASES', '24'))
OFFSET = int(os.environ.get('MZ_FUZZ_SEED_OFFSET', '0'))  # soak runs: another region of every case generator's seed space


def _draw_case(i):
    rs = np.random.RandomState(9000 + i + 100000 * OFFSET)
    tuned = rs.rand() < 0.6
    if tuned:  # shapes the launcher routes to k_search_fast
        P, H = int(rs.choice([256, 512])), 64
        A = int(rs.choice([2, 2, 4, 10, 3, 7, 16]))
        sup = [(31, 31), (1, 1), (21, 31), (31, 7), (1, 31)][rs.randint(5)]
    else:
        P, H = int(rs.choice([16, 40, 96, 128, 200, 384])), int(rs.choice([8, 20, 32, 64, 100]))
        A = int(rs.randint(2, 19)) if rs.rand() < 0.8 else int(rs.choice([30, 64, 130, 226]))  # (wide action sets: other tree layouts)
        sup = (int(rs.choice([1, 5, 31, 61])), int(rs.choice([1, 9, 31])))
    ishape = [(4, 5), (3, 3, 3), (7,), (2, 2, 2), (4, 9)][rs.randint(5)]
    board = bool(rs.rand() < 0.4)
    bounds = [(-1.0, 1.0), (0.0, 1.0), (-3.5, 7.25)][rs.randint(3)] if (board or rs.rand() < 0.3) else None  # (MinMaxStats' known bounds, mcts.py:36-38)
    S = int(rs.choice([1, 2, 5, 17, 25, 50, 60, 120, 300]))  # (long searches leave the LDS-resident tree layouts: other kernels)
    B = int(rs.choice([1, 3, 16, 17, 40, 70]))
    return dict(case=(f'fuzz{i}', ishape, A, P, sup[0], sup[1], H, 500 + i), board=board, bounds=bounds, S=S, B=B,
                discount=1.0 if board else float(rs.choice([0.997, 0.9, 1.0])), alpha=float(rs.choice([0.03, 0.25, 1.0])),
                eps=float(rs.choice([0.25, 0.0, 0.5])), seed=int(rs.randint(1 << 30)))


@pytest.mark.parametrize('i', range(CASES))
def test_random_configuration_bit_exact_vs_oracle(oracle, i):
    c = _draw_case(i)
    case, board, S, B = c['case'], c['board'], c['S'], c['B']
    A = case[2]
    net = build_mlp(case)
    onet = _oracle_net(oracle, net, 'mlp')
    kw = dict(num_simulations=S, discount=c['discount'], is_board_game=board, known_bounds=c['bounds'], root_dirichlet_alpha=c['alpha'],
              root_exploration_eps=c['eps'])
    ocfg = oracle.make_config(A, S, c['discount'], board, c['bounds'], c['alpha'], c['eps'])
    p = _planner(net, B, **kw)
    rs = np.random.RandomState(c['seed'])
    obs = rs.uniform(-1, 1, size=(B,) + tuple(case[1])).astype(np.float32)
    mask = rs.rand(B, A) < 0.7
    mask[np.arange(B), rs.randint(0, A, B)] = True
    cur = rs.randint(1, 3, B).astype(np.int32) if board else np.ones(B, np.int32)
    opp = (3 - cur).astype(np.int32) if board else np.ones(B, np.int32)
    temp = rs.choice([1.0, 0.5, 0.25, 0.1, 0.0], size=B)
    noise = rs.dirichlet(np.full(A, max(c['alpha'], 0.05)), size=B)
    u_tie = rs.rand(B, 4 * S + 8)
    u_final = rs.rand(B)
    for det in (False, True):
        r = p.search(obs, mask, cur, opp, temp, det, noise=None if det else noise, u_tie=u_tie, u_final=u_final)
        o = oracle.uct_search_batch(ocfg, onet, obs, mask.astype(np.uint8), cur, opp, temp, det, noise=None if det else noise, u_tie=u_tie,
                                    u_final=u_final)
        np.testing.assert_array_equal(r['visits'], o['visits'], err_msg=str(c))
        np.testing.assert_array_equal(r['pi'], o['pi'], err_msg=str(c))
        np.testing.assert_array_equal(r['action'], o['action'], err_msg=str(c))
        np.testing.assert_array_equal(r['root_value'], o['root_value'], err_msg=str(c))
    if B > 1:  # a smaller batch on the same planner (fewer envs than it was created for): the same rows again
        b = int(rs.randint(1, B))
        r2 = p.search(obs[:b], mask[:b], cur[:b], opp[:b], temp[:b], True, noise=None, u_tie=u_tie[:b], u_final=u_final[:b])
        for key in ('visits', 'pi', 'action', 'root_value'):
            np.testing.assert_array_equal(r2[key], r[key][:b], err_msg=str((c, 'sub-batch', b)))
    p.close()


CONV_CASES_N = int(os.environ.get('MZ_FUZZ_CONV_CASES', '10'))


def _draw_conv_case(i):
    rs = np.random.RandomState(7000 + i + 100000 * OFFSET)
    N = int(rs.choice([3, 4, 5, 6, 7, 8, 9, 11]))
    planes = int(rs.choice([8, 16, 16, 24, 32, 48]))
    blocks = int(rs.choice([1, 1, 2]))
    chans = int(rs.choice([3, 5, 9]))
    # keep the scalar oracle's work per case bounded (~1 G MAC)
    per_sim = planes * planes * 9 * N * N * (2 + 4 * blocks)
    budget = 1.2e9
    S = int(rs.choice([2, 5, 9, 14]))
    B = int(rs.choice([1, 4, 9, 18, 33]))
    while per_sim * S * B > budget and B > 1:
        B = max(1, B // 2)
    while per_sim * S * B > budget and S > 2:
        S -= 1
    return dict(case=(f'cfuzz{i}', 'board', (chans, N, N), N * N + 1, blocks, planes, 1, 1, 800 + i), S=S, B=B,
                alpha=float(rs.choice([0.03, 0.25])), seed=int(rs.randint(1 << 30)))


@pytest.mark.parametrize('i', range(CONV_CASES_N))
def test_random_board_conv_configuration_bit_exact_vs_oracle(oracle, i):
    """The conv path's kernels are chosen by geometry (images per workgroup, pixel tiles, fused residual tower or one launch per conv, the
    sparse action terms fused into the epilogue or added by their own kernel, LDS- or HBM-resident trees): random board sizes, channel
    counts, tower depths and batch sizes, every env bit-exact against the oracle."""
    from helpers import build_conv

    c = _draw_conv_case(i)
    case, S, B = c['case'], c['S'], c['B']
    A = case[3]
    net = build_conv(case)
    onet = _oracle_net(oracle, net, 'conv')
    kw = dict(num_simulations=S, discount=1.0, is_board_game=True, known_bounds=(-1.0, 1.0), root_dirichlet_alpha=c['alpha'], root_exploration_eps=0.25)
    ocfg = oracle.make_config(A, S, 1.0, True, (-1.0, 1.0), c['alpha'], 0.25)
    p = _planner(net, B, **kw)
    rs = np.random.RandomState(c['seed'])
    obs = rs.randint(0, 2, size=(B,) + tuple(case[2])).astype(np.float32)
    mask = rs.rand(B, A) < 0.8
    mask[np.arange(B), rs.randint(0, A, B)] = True
    cur = rs.randint(1, 3, B).astype(np.int32)
    opp = (3 - cur).astype(np.int32)
    temp = rs.choice([1.0, 0.5, 0.0], size=B)
    noise = rs.dirichlet(np.full(A, 0.25), size=B)
    u_tie = rs.rand(B, 4 * S + 8)
    u_final = rs.rand(B)
    for det in ((False, True) if i % 2 else (False,)):  # (every other case also in deterministic mode: float32 prior path, argmax play)
        r = p.search(obs, mask, cur, opp, temp, det, noise=None if det else noise, u_tie=u_tie, u_final=u_final)
        o = oracle.uct_search_batch(ocfg, onet, obs, mask.astype(np.uint8), cur, opp, temp, det, noise=None if det else noise, u_tie=u_tie,
                                    u_final=u_final)
        np.testing.assert_array_equal(r['visits'], o['visits'], err_msg=str(c))
        np.testing.assert_array_equal(r['pi'], o['pi'], err_msg=str(c))
        np.testing.assert_array_equal(r['action'], o['action'], err_msg=str(c))
        np.testing.assert_array_equal(r['root_value'], o['root_value'], err_msg=str(c))
    p.close()


LEARN_CASES_N = int(os.environ.get('MZ_FUZZ_LEARN_CASES', '10'))
# the learner comparisons below accept a looser bar next to a detected kink of the loss.  So that a real regression cannot hide
# there, the cases that have ever NEEDED the looser bar are listed by generator index (seed offset 0, indices < 300: what tools/dev/deep_parity.sh
# runs): a case outside the list that needs it fails.  MZ_FUZZ_RECORD_CARVED=<file>: append "kind index" lines instead (to rebuild the list: which
# of two branches a float32 pass takes at a kink depends on the kernels' summation orders, so a change of tiling moves a few cases in or out).
KNOWN_CARVED = {'mlp': set(),  # (none in 300)
                'convkf': set(),  # kink-free weights (none in 300)
                'conv': set()}  # random weights: since round 6 compared on the HIP pass's own branches (tests/forced_masks.py): no carve-out (rounds 4-5: 103 of 300)


def _carved(kind, i, detail):
    rec = os.environ.get('MZ_FUZZ_RECORD_CARVED')
    if rec:
        with open(rec, 'a') as f:
            f.write(f'{kind} {i} {detail}\n')
        return
    if OFFSET == 0 and i < 300:
        assert i in KNOWN_CARVED[kind], (kind, i, 'needs the kink tolerance but is not a known carved-out case', detail)


def _draw_learn_case(i):
    rs = np.random.RandomState(5000 + i + 100000 * OFFSET)
    if rs.rand() < 0.55:  # shapes with register-resident builds (512 / 256 planes, hidden 64)
        P, H = int(rs.choice([512, 256])), 64
        A = int(rs.choice([2, 4, 10, 6]))
        sup = [(31, 31), (1, 1), (31, 1), (21, 31)][rs.randint(4)]
        ishape = [(4, 5), (9, 3, 3), (4, 9)][rs.randint(3)]
    else:         # anything else: the generic builds
        P, H = int(rs.choice([32, 40, 96, 160])), int(rs.choice([16, 20, 32, 64]))
        A = int(rs.randint(2, 12))
        sup = (int(rs.choice([1, 7, 31])), int(rs.choice([1, 5, 31])))
        ishape = [(7,), (3, 4), (2, 2, 2), (5, 5)][rs.randint(4)]
    # batch sizes across the launcher's regimes: plane-sliced stages, persistent chains (>= 96 tiles), streaming heads (>= 16 tiles), split reductions
    B = int(rs.choice([3, 16, 33, 128, 250, 640, 1600, 2100, 4100]))
    return dict(case=(f'lfuzz{i}', ishape, A, P, sup[0], sup[1], H, 900 + i), B=B, seed=int(rs.randint(1 << 30)), weights=bool(rs.rand() < 0.5),
                K=int(rs.choice([5, 5, 1, 2, 3, 8])),  # (unroll length: 5 in every reference configuration)
                opt=dict(lr=float(rs.choice([1e-3, 2e-2, 3e-4])), betas=(float(rs.choice([0.9, 0.8])), float(rs.choice([0.999, 0.95]))),
                         eps=float(rs.choice([1e-8, 1e-6])), wd=float(rs.choice([0.0, 1e-4, 1e-2])), clip=float(rs.choice([0.0, 0.5, 40.0]))))


@pytest.mark.parametrize('i', range(LEARN_CASES_N))
def test_random_learner_configuration_matches_autograd(i):
    """The HIP learner step over its launcher's regimes and the nets' shape space against PyTorch-ROCm autograd on the same batch: loss to
    2e-4, every gradient tensor to 3e-3 of its norm, priorities to 2e-3.  Batches that sit within rounding of a kink of the loss (detected on
    the autograd side, see below: once in ~7 000 cases) are only held to 5e-2."""
    import torch

    from muzero_amd import learner
    from muzero_amd.hip_learner import HipLearner
    from muzero_amd.replay import Transition

    c = _draw_learn_case(i)
    case, B = c['case'], c['B']
    A, K = case[2], c['K']
    dev = torch.device('cuda', 0)
    net_a = build_mlp(case).to(dev)
    import copy
    net_b = copy.deepcopy(net_a)
    net_a.train()
    rs = np.random.RandomState(c['seed'])
    tr = Transition(rs.uniform(-1, 1, (B,) + tuple(case[1])).astype(np.float32), rs.randint(0, A, (B, K)).astype(np.int8),
                    rs.dirichlet(np.ones(A), size=(B, K)).astype(np.float32), rs.uniform(-3, 3, (B, K)).astype(np.float32),
                    rs.uniform(-1, 1, (B, K)).astype(np.float32))
    w = rs.uniform(0.3, 1.0, B).astype(np.float32) if c['weights'] else np.ones(B, np.float32)
    # non-smooth points of the loss: a ReLU pre-activation within rounding of zero, or two features of an un-normalised state within rounding
    # of each other at its minimum / maximum (util.py:31-36: the min / max gradient goes to ONE index).  There the two implementations'
    # summation orders may pick different branches, and everything upstream of that sample moves by its contribution.
    margins = []

    def watch_relu(_m, _i, out):
        margins.append(float(out.detach().abs().min()))

    def watch_state(_m, _i, out):
        v = out.detach().sort(dim=1).values
        rng = (v[:, -1] - v[:, 0]).clamp_min(1e-30)
        margins.append(float(torch.minimum((v[:, 1] - v[:, 0]) / rng, (v[:, -1] - v[:, -2]) / rng).min()))

    hooks = []
    for name, m in net_a.named_modules():
        if isinstance(m, torch.nn.Linear) and name.endswith('.0'):
            hooks.append(m.register_forward_hook(watch_relu))
        if name in ('represent_net.net.2', 'dynamics_net.transition_net.2'):
            hooks.append(m.register_forward_hook(watch_state))
    la, pa = learner.calc_loss(net_a, dev, tr, torch.from_numpy(w).to(dev))
    la.backward()
    for hk in hooks:
        hk.remove()
    smooth = min(margins) > 5e-6
    hl = HipLearner(net_b, dev, K, B, lr=1e-3)
    ring = {f: torch.from_numpy(np.ascontiguousarray(getattr(tr, f))).to(dev) for f in Transition._fields}
    ring['state'] = ring['state'].reshape(B, -1).contiguous()
    lb, pb = hl.grad(ring, None, torch.from_numpy(w).to(dev), B)
    la_f = float(la.detach())
    assert abs(la_f - float(lb)) <= 2e-4 * max(1.0, abs(la_f)), c
    np.testing.assert_allclose(pb.cpu().numpy(), pa, rtol=2e-3, atol=2e-3, err_msg=str(c))
    for k, p_ in net_a.named_parameters():
        a, b = p_.grad.detach().cpu().numpy().ravel().astype(np.float64), hl.grad_views[k].cpu().numpy().ravel().astype(np.float64)
        na, err = float(np.linalg.norm(a)), float(np.linalg.norm(a - b))
        if err <= 3e-3 * max(na, 1e-7) + 2e-7:  # (+ float32 rounding of the sums themselves: a two-action policy bias gradient cancels to ~5e-6)
            continue  # (the normal case, kink nearby or not)
        assert not smooth and err <= 5e-2 * max(na, 1e-7), (k, c, err, na, 'smooth' if smooth else 'near a kink', min(margins))
        _carved('mlp', i, f'{k} {err / max(na, 1e-7):.2e}')
    # one optimizer step with the drawn hyper-parameters against torch.optim.Adam + clip_grad_norm_ -- on IDENTICAL gradients (the kernels'
    # gradient copied into the torch parameters: gradient parity is the comparison above; with each side on its own gradient an entry next
    # to zero takes -lr sign(g) with opposite signs, seen twice in 6 000 cases), so this checks the clip / Adam / weight-decay arithmetic alone
    hp = c['opt']
    pb = dict(net_b.named_parameters())
    before_b = {k: v.detach().clone() for k, v in pb.items()}
    before_a = {k: v.detach().clone() for k, v in net_a.named_parameters()}
    for k, p_ in net_a.named_parameters():
        p_.grad.copy_(hl.grad_views[k])
    opt = torch.optim.Adam(net_a.parameters(), lr=hp['lr'], betas=hp['betas'], eps=hp['eps'], weight_decay=hp['wd'])
    if hp['clip'] > 0:
        torch.nn.utils.clip_grad_norm_(net_a.parameters(), hp['clip'])
    gclip = {k: p_.grad.detach().clone() for k, p_ in net_a.named_parameters()}
    opt.step()
    hl.lr_init, hl.betas, hl.eps, hl.weight_decay, hl.max_grad_norm = hp['lr'], hp['betas'], hp['eps'], hp['wd'], hp['clip']
    hl.apply(clip=hp['clip'] > 0)
    for k, p_ in net_a.named_parameters():
        da, db = (p_.detach() - before_a[k]).cpu().numpy(), (pb[k].detach() - before_b[k]).cpu().numpy()
        ulp = 2.4e-7 * max(1.0, float(before_a[k].abs().max()))  # (the updated weight is rounded to float32: one ulp of the weight either way)
        # Adam's first update is -lr g' / (|g'| + eps) with g' = clipped gradient + wd w: where the two terms cancel (seen: 2.53265e-4 -
        # 2.53247e-4 against eps 1e-8) the rounding of g' itself is amplified by lr eps / (|g'| + eps)^2
        gc, ww = gclip[k].cpu().numpy().astype(np.float64), hp['wd'] * before_a[k].cpu().numpy().astype(np.float64)
        amp = hp['lr'] * hp['eps'] / (np.abs(gc + ww) + hp['eps']) ** 2 * 4e-7 * (np.abs(gc) + np.abs(ww))
        excess = np.abs(da - db) - (2e-4 * hp['lr'] + ulp + amp)
        assert float(excess.max()) <= 0.0, (k, c, float(np.abs(da - db).max()))
    hl.close()


SELFPLAY_CASES_N = int(os.environ.get('MZ_FUZZ_SELFPLAY_CASES', '6'))


@pytest.mark.parametrize('i', range(SELFPLAY_CASES_N))
def test_random_net_in_device_selfplay_equals_oracle_search(oracle, i):
    """Device self-play (production Philox draws, env step fused into the search kernel where a build exists) with nets drawn over the
    shape space that fits the device envs: every move's policy, root value and action equal the oracle's search on the recorded inputs
    and the captured draws (the comparison of test_gpu_selfplay.py::test_selfplay_search_outputs_equal_oracle_search)."""
    from test_gpu_selfplay import selfplay_search_vs_oracle

    rs = np.random.RandomState(3000 + i + 100000 * OFFSET)
    game = ['cartpole', 'tictactoe'][rs.randint(2)]
    ishape, A = ((4, 5), 2) if game == 'cartpole' else ((9, 3, 3), 10)
    P, H = int(rs.choice([32, 96, 256, 512])), int(rs.choice([16, 32, 64, 64]))
    sup = [(31, 31), (1, 1), (11, 31), (31, 1)][rs.randint(4)]
    case = (f'sfuzz{i}', ishape, A, P, sup[0], sup[1], H, 1300 + i)
    S, B = int(rs.choice([5, 25, 50])), int(rs.choice([16, 48, 100]))
    reload_case = (case[0] + 'r',) + tuple(case[1:7]) + (case[7] + 5000,) if rs.rand() < 0.5 else None  # the same net with other weights
    selfplay_search_vs_oracle(oracle, game, case, S, B, 12, seed=int(rs.randint(1 << 20)), expect_resets=False, reload_case=reload_case)


EPI_CASES_N = int(os.environ.get('MZ_FUZZ_EPILOGUE_CASES', '6'))


@pytest.mark.parametrize('i', range(EPI_CASES_N))
def test_random_epilogue_configuration_matches_host_assembler(i):
    """The device epilogue (targets, priorities, unroll windows written into the HBM replay ring) over its parameter space -- unroll length,
    mid-episode flush length, n-step horizon, env count, chunking of the moves, ring capacities that wrap -- against the host assembler on
    the same records, exact."""
    import types

    from test_gpu_epilogue import _compare, _run

    rs = np.random.RandomState(2000 + i + 100000 * OFFSET)
    game = ['cartpole', 'tictactoe'][rs.randint(2)]
    board = game == 'tictactoe'
    K = int(rs.choice([1, 2, 3, 5, 7]))
    cfg = types.SimpleNamespace(is_board_game=board, unroll_steps=K, discount=1.0 if board else float(rs.choice([0.997, 0.9])),
                                acc_seq_length=200 if board else int(rs.choice([3, 6, 20, 200])), td_steps=0 if board else int(rs.choice([1, 3, 10])))
    B, chunk = int(rs.choice([5, 16, 33, 64])), int(rs.choice([1, 4, 8, 16]))
    moves = chunk * max(int(rs.choice([4, 8, 12])), -(-64 // chunk))  # (at least 64 moves: random CartPole episodes last ~20 steps)
    capacity = int(rs.choice([64, 256, 8192]))
    p, rp, origin, host, n = _run(game, B, moves, chunk, cfg, capacity=capacity, seed=int(rs.randint(1 << 20)))
    assert _compare(rp, origin, host, n) == min(n, capacity)
    p.close()


GOMOKU_CASES_N = int(os.environ.get('MZ_FUZZ_GOMOKU_CASES', '4'))


@pytest.mark.parametrize('i', range(GOMOKU_CASES_N))
def test_random_conv_net_in_device_gomoku_selfplay_equals_oracle_search(oracle, i):
    """The same for the conv path: device Gomoku (N x N, five in a row, stacked history) with board nets drawn over sizes, plane counts and
    tower depths -- HBM-resident trees, fused towers or per-conv launches, fused action terms -- every move of device self-play equal to the
    oracle's search on the recorded board and the captured draws."""
    from test_gpu_selfplay import selfplay_search_vs_oracle

    rs = np.random.RandomState(4000 + i + 100000 * OFFSET)
    N = int(rs.choice([5, 6, 7, 9]))
    stack = 4 + 0 * int(rs.choice([1, 2, 4]))  # (the device Gomoku env is the reference configuration: 4 stacked positions, 9 planes)
    planes, blocks = int(rs.choice([8, 16, 32])), int(rs.choice([1, 2]))
    case = (f'gfuzz{i}', 'board', (2 * stack + 1, N, N), N * N + 1, blocks, planes, 1, 1, 1500 + i)
    S, B = int(rs.choice([3, 8])), int(rs.choice([4, 12]))
    reload_case = (case[0] + 'r',) + tuple(case[1:8]) + (case[8] + 5000,) if rs.rand() < 0.5 else None
    selfplay_search_vs_oracle(oracle, 'gomoku', case, S, B, 10, seed=int(rs.randint(1 << 20)), expect_resets=False, reload_case=reload_case)


ATARI_CASES_N = int(os.environ.get('MZ_FUZZ_ATARI_CASES', '4'))


@pytest.mark.parametrize('i', range(ATARI_CASES_N))
def test_random_atari_conv_configuration_bit_exact_vs_oracle(oracle, i):
    """Atari-shaped nets (96 x 96 frames through the stride-2 / pooled representation down to 6 x 6, categorical heads) over frame
    counts, plane counts, tower depths, action counts and support sizes: batched search bit-exact against the oracle."""
    from helpers import build_conv

    rs = np.random.RandomState(6000 + i + 100000 * OFFSET)
    frames, planes, blocks = int(rs.choice([2, 4, 8])), int(rs.choice([8, 16, 32])), int(rs.choice([1, 2]))
    A, sup = int(rs.randint(3, 19)), [(11, 11), (31, 61), (61, 31), (5, 21)][rs.randint(4)]
    case = (f'afuzz{i}', 'atari', (frames, 96, 96), A, blocks, planes, sup[0], sup[1], 1700 + i)
    S, B = int(rs.choice([2, 6, 10])), int(rs.choice([1, 3, 6]))
    net = build_conv(case)
    onet = _oracle_net(oracle, net, 'conv')
    kw = dict(num_simulations=S, discount=0.997, is_board_game=False, known_bounds=None, root_dirichlet_alpha=0.25, root_exploration_eps=0.25)
    ocfg = oracle.make_config(A, S, 0.997, False, None, 0.25, 0.25)
    p = _planner(net, B, **kw)
    obs = rs.uniform(0, 1, size=(B, frames, 96, 96)).astype(np.float32)
    mask = np.ones((B, A), bool)
    cur = opp = np.ones(B, np.int32)
    temp = rs.choice([1.0, 0.5, 0.0], size=B)
    noise = rs.dirichlet(np.full(A, 0.25), size=B)
    u_tie = rs.rand(B, 4 * S + 8)
    u_final = rs.rand(B)
    for det in (False, True):
        r = p.search(obs, mask, cur, opp, temp, det, noise=None if det else noise, u_tie=u_tie, u_final=u_final)
        o = oracle.uct_search_batch(ocfg, onet, obs, mask.astype(np.uint8), cur, opp, temp, det, noise=None if det else noise, u_tie=u_tie,
                                    u_final=u_final)
        np.testing.assert_array_equal(r['visits'], o['visits'], err_msg=str(case))
        np.testing.assert_array_equal(r['pi'], o['pi'], err_msg=str(case))
        np.testing.assert_array_equal(r['action'], o['action'], err_msg=str(case))
        np.testing.assert_array_equal(r['root_value'], o['root_value'], err_msg=str(case))
    p.close()


PLUMB_CASES_N = int(os.environ.get('MZ_FUZZ_PLUMBING_CASES', '6'))


@pytest.mark.parametrize('i', range(PLUMB_CASES_N))
def test_random_learner_batch_plumbing_is_exact(i):
    """One learner serving a SEQUENCE of batches of different sizes (below its capacity), gathered from a larger replay ring by index vectors
    with repeats, float32 or int8 states, int8 or int16 action fields, with and without importance weights -- against a fresh learner of
    exactly that batch size on the same rows stacked contiguously: loss, priorities and every gradient bit for bit (the job tables follow
    the batch size, stale gradient slices are cleared, samples past the batch contribute nothing)."""
    import copy

    import torch

    from muzero_amd.hip_learner import HipLearner
    from muzero_amd.replay import Transition

    rs = np.random.RandomState(1000 + i + 100000 * OFFSET)
    P, H = (int(rs.choice([512, 256])), 64) if rs.rand() < 0.6 else (int(rs.choice([32, 96])), int(rs.choice([16, 32])))
    A = int(rs.choice([2, 4, 10, 200])) if P >= 256 else int(rs.randint(2, 9))
    A = A if A <= 32 else (A if P < 256 else 10)  # (the tuned builds cover <= 32 actions; 200 actions exercise int16 fields on the generic builds)
    if P < 256 and rs.rand() < 0.3:
        A = 200
    sup = [(31, 31), (1, 1), (11, 31)][rs.randint(3)]
    ishape = [(4, 5), (9, 3, 3), (7,)][rs.randint(3)]
    case = (f'pfuzz{i}', ishape, A, P, sup[0], sup[1], H, 2100 + i)
    K = int(rs.choice([5, 5, 2, 7]))
    int8_state, act16 = bool(rs.rand() < 0.3), A > 128
    dev = torch.device('cuda', 0)
    net = build_mlp(case).to(dev)
    cap, Bmax = int(rs.choice([300, 1000])), int(rs.choice([64, 200, 700]))
    slices = int(rs.choice([1, 1, 3]))
    st = rs.randint(0, 2, (cap,) + tuple(ishape)).astype(np.int8) if int8_state else rs.uniform(-1, 1, (cap,) + tuple(ishape)).astype(np.float32)
    ring = dict(state=torch.from_numpy(st).to(dev).reshape(cap, -1).contiguous(),
                action=torch.from_numpy(rs.randint(0, A, (cap, K)).astype(np.int16 if act16 else np.int8)).to(dev),
                pi_prob=torch.from_numpy(rs.dirichlet(np.ones(A), size=(cap, K)).astype(np.float32)).to(dev),
                value=torch.from_numpy(rs.uniform(-3, 3, (cap, K)).astype(np.float32)).to(dev),
                reward=torch.from_numpy(rs.uniform(-1, 1, (cap, K)).astype(np.float32)).to(dev))
    big = HipLearner(copy.deepcopy(net), dev, K, Bmax, lr=1e-3, grad_slices=slices)
    for step in range(3):
        b = int(rs.randint(1, Bmax + 1))
        idx = torch.from_numpy(rs.randint(0, cap, b).astype(np.int64)).to(dev)
        w = torch.from_numpy(rs.uniform(0.3, 1.0, b).astype(np.float32)).to(dev) if rs.rand() < 0.5 else None
        loss, prio = big.grad(ring, idx, w, b)
        got = (float(loss), prio.cpu().numpy().copy(), big.grad_flat.cpu().numpy().copy())
        ref = HipLearner(copy.deepcopy(net), dev, K, b, lr=1e-3, grad_slices=slices)
        stacked = {f: t.index_select(0, idx).contiguous() for f, t in ring.items()}
        l2, p2 = ref.grad(stacked, None, w, b)
        assert got[0] == float(l2), (case, step, b)
        np.testing.assert_array_equal(got[1], p2.cpu().numpy(), err_msg=str((case, step, b)))
        np.testing.assert_array_equal(got[2], ref.grad_flat.cpu().numpy(), err_msg=str((case, step, b)))
        ref.close()
    big.close()


INFER_CASES_N = int(os.environ.get('MZ_FUZZ_INFER_CASES', '8'))


@pytest.mark.parametrize('i', range(INFER_CASES_N))
def test_random_net_inference_api_bit_exact_vs_oracle(oracle, i):
    """`initial_inference` / `recurrent_inference` (network.py:62-111) as the planner's batched API serves them -- their own kernels and dense
    output paths, not the search's -- for nets drawn over the MLP and board-conv shape spaces and ragged batch sizes: hidden states, policies,
    rewards and values bit-exact against the oracle's networks."""
    rs = np.random.RandomState(8000 + i + 100000 * OFFSET)
    if rs.rand() < 0.5:
        P, H = int(rs.choice([16, 40, 96, 256, 512])), int(rs.choice([8, 20, 64]))
        A, sup = int(rs.randint(2, 19)), (int(rs.choice([1, 5, 31, 61])), int(rs.choice([1, 9, 31])))
        ishape = [(4, 5), (3, 3, 3), (7,), (4, 9)][rs.randint(4)]
        case, kind = (f'ifuzz{i}', ishape, A, P, sup[0], sup[1], H, 2500 + i), 'mlp'
        net = build_mlp(case)
        obs_shape = ishape
    else:
        from helpers import build_conv

        N, planes, blocks = int(rs.choice([3, 5, 6, 7, 9, 11])), int(rs.choice([8, 16, 32, 48])), int(rs.choice([1, 2]))
        chans = int(rs.choice([3, 5, 9]))
        A = N * N + 1
        case, kind = (f'ifuzz{i}', 'board', (chans, N, N), A, blocks, planes, 1, 1, 2500 + i), 'conv'
        net = build_conv(case)
        obs_shape = (chans, N, N)
    B = int(rs.choice([1, 5, 17, 37]))
    onet = _oracle_net(oracle, net, kind)
    p = _planner(net, 64)
    obs = (rs.randint(0, 2, size=(B,) + tuple(obs_shape)) if kind == 'conv' else rs.uniform(-1, 1, size=(B,) + tuple(obs_shape))).astype(np.float32)
    hidden, pi, value = p.initial_inference(obs)
    actions = rs.randint(0, A, size=B).astype(np.int32)
    h2, reward, pi2, value2 = p.recurrent_inference(hidden, actions)
    for b in range(B):
        oh, _, opi, ov = onet.initial_inference(obs[b])
        np.testing.assert_array_equal(np.asarray(hidden[b]).reshape(-1), np.asarray(oh).reshape(-1), err_msg=str(case))
        np.testing.assert_array_equal(pi[b], opi, err_msg=str(case))
        assert value[b] == np.float32(ov), case
        oh2, orw, opi2, ov2 = onet.recurrent_inference(oh, int(actions[b]))
        np.testing.assert_array_equal(np.asarray(h2[b]).reshape(-1), np.asarray(oh2).reshape(-1), err_msg=str(case))
        np.testing.assert_array_equal(pi2[b], opi2, err_msg=str(case))
        assert reward[b] == np.float32(orw) and value2[b] == np.float32(ov2), case
    p.close()


CONV_LEARN_CASES_N = int(os.environ.get('MZ_FUZZ_CONV_LEARN_CASES', '8'))


def _draw_conv_learn_case(i):
    rs = np.random.RandomState(7700 + i + 100000 * OFFSET)
    board = int(rs.choice([3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]))
    planes = int(rs.choice([8, 16, 24, 32, 40, 64, 128]))
    budget = 2_500_000  # elements of one activation tensor x layers: keeps a case (and its float64 autograd twin) under a second
    blocks = int(rs.choice([1, 1, 2, 3]))
    K = int(rs.choice([5, 5, 1, 2, 3, 6]))
    B = int(rs.choice([1, 2, 5, 9, 16, 23, 40]))
    while B > 1 and B * planes * board * board * (1 + 2 * blocks) * (1 + 2 * K) > budget * 8:
        B = max(1, B // 2)
    return dict(board=board, planes=planes, blocks=blocks, chan=int(rs.randint(1, 10)), K=K, B=B, int8=bool(rs.rand() < 0.5), weights=bool(rs.rand() < 0.6),
                seed=int(rs.randint(1 << 30)), opt=dict(lr=float(rs.choice([1e-3, 2e-2])), wd=float(rs.choice([0.0, 1e-4])), clip=float(rs.choice([0.0, 0.5, 40.0]))))


@pytest.mark.parametrize('i', range(CONV_LEARN_CASES_N))
def test_random_conv_learner_configuration_matches_float64_autograd(i):
    """Round 5: the conv learner's kernels (csrc/mz_learn_conv.h) over board sizes 3-15 (every pixel tiling, images per workgroup, pitch layout of the
    weight gradient), plane counts on and off the 16-channel tile, 1-3 blocks, unroll 1-6, ragged batches, int8 / float states, int8 / int16 actions,
    with / without importance weights: loss, priorities, every gradient and the BatchNorm running statistics against float64 PyTorch-ROCm autograd; then
    one optimizer step with drawn Adam / clip settings against torch.optim.Adam.  Gradients are compared with float64 autograd on the branches the
    HIP pass took (tests/forced_masks.py): 1e-4 of each tensor's largest entry, or 4 x PyTorch-ROCm's own float32 error on that branch where a batch is
    ill-conditioned (batch 1, a near-constant plane under BatchNorm) -- rounds 4-5 held a third of these cases to 0.25 and kept a list of them."""
    import copy

    import torch

    from muzero_amd import learner
    from muzero_amd.hip_learner import HipLearner
    from test_gpu_conv_learner import _batch, _f64_reference, _net, _ring

    c = _draw_conv_learn_case(i)
    dev = torch.device('cuda', 0)
    net, A = _net(c['board'], c['planes'], c['blocks'], c['chan'], 4000 + i, dev)
    net.train()
    rs = np.random.RandomState(c['seed'])
    B, K, shape = c['B'], c['K'], (c['chan'], c['board'], c['board'])
    tr = _batch(rs, B, shape, A, K=K, int8_state=c['int8'])
    w = rs.uniform(0.3, 1.0, B).astype(np.float32) if c['weights'] else np.ones(B, np.float32)
    loss_d, prio_d, gd, sd_d, closest = _f64_reference(net, tr._replace(state=tr.state.astype(np.float64)), w, dev)
    o = c['opt']
    hl = HipLearner(net, dev, K, B + int(rs.randint(0, 5)), lr=o['lr'], weight_decay=o['wd'], clip_grad=o['clip'] > 0, max_grad_norm=o['clip'] or 40.0)
    loss, prio = hl.grad(_ring(tr, dev), None, torch.from_numpy(w).to(dev) if c['weights'] else None, B)
    assert abs(float(loss) - loss_d) <= 1e-4 * max(1.0, abs(loss_d)), c
    np.testing.assert_allclose(prio.cpu().numpy(), prio_d.cpu().numpy(), rtol=1e-3, atol=1e-4, err_msg=str(c))
    # gradients: float64 autograd on the branches THIS pass took (round 6, tests/forced_masks.py): no kink allowance, no list of carved-out cases
    from test_gpu_conv_learner import _same_branch, same_branch_bar

    errs, err32, loss_f, _, flipped, gd = _same_branch(hl, net, tr, w, B, K, dev, want_grads=True)
    wk, worst, bar = same_branch_bar(errs, err32)
    assert worst <= bar and bar <= 0.1, (c, wk, worst, bar, flipped, closest)
    sd = net.state_dict()
    for k, v in sd_d.items():
        if 'running' in k:
            assert float((v - sd[k].double()).abs().max()) <= 3e-5 * max(1.0, float(v.abs().max())), (c, k)
        if 'num_batches_tracked' in k:
            assert int(v) == int(sd[k]), (c, k)
    # one Adam step from the (same-branch) float64 gradient with torch's own optimizer on a float64 twin of the weights
    twin = copy.deepcopy(net).double()
    for (k, p) in twin.named_parameters():
        p.grad = gd[k].clone()
    opt = torch.optim.Adam(twin.parameters(), lr=o['lr'], weight_decay=o['wd'])
    if o['clip'] > 0:
        torch.nn.utils.clip_grad_norm_(twin.parameters(), o['clip'])
    opt.step()
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    hl.apply()
    for (k, p), (_, q) in zip(net.named_parameters(), twin.named_parameters()):
        moved = float((q - before[k].double()).abs().max())
        # Adam's first step moves every weight by ~lr whatever the gradient's size: elements whose gradient is at rounding distance from 0 may go
        # the other way in float32 -- a mean bar for the tensor, and a 2 lr bar for single elements
        d = (p.double() - q).abs()
        assert float(d.max()) <= 2.2 * o['lr'] + 1e-6, (c, k, float(d.max()), moved)
        # (the mean bar over the elements whose gradient has a SIGN in float32: a tensor whose exact gradient cancels -- the BatchNorm shift in
        # front of a train-mode BatchNorm, entries 1e-8 of their neighbours -- steps by +- lr at random in any float32 pass; rounds 4-5 never got
        # here on the third of the cases they called kinked)
        part_max = max(float(g.abs().max()) for kk, g in gd.items() if kk.split('.')[0] == k.split('.')[0])
        # (the gradients agree to 1e-4 of the tensor's largest entry: an element below ~1e-3 of it may carry either sign in float32)
        sure = gd[k].abs() > 1e-3 * max(float(gd[k].abs().max()), 1e-2 * part_max)
        if bool(sure.any()):
            assert float(d[sure].mean()) <= 0.02 * max(moved, 1e-12) + 1e-7, (c, k, float(d[sure].mean()), moved, int(sure.sum()), sure.numel())


@pytest.mark.parametrize('i', range(CONV_LEARN_CASES_N))
def test_random_conv_learner_configuration_kink_free(i):
    """The same generator with KINK-FREE weights (tests/test_gpu_atari_learner.kinkfree_state_dict: every ReLU channel on or off for the whole batch, a
    mix of both; confirmed by the float64 probe): no carve-out -- every gradient tensor within 2e-3 of float64 autograd (a third of the random-weight
    cases above sit within rounding of a ReLU boundary somewhere and are held to 8e-2 only)."""
    import torch

    from muzero_amd.hip_learner import HipLearner
    from muzero_amd.network import MuZeroBoardGameNet
    from test_gpu_atari_learner import f32_errors, grad_errors, kinkfree_state_dict, kinkfree_worst
    from test_gpu_conv_learner import _batch, _f64_reference, _ring

    c = _draw_conv_learn_case(i)
    dev = torch.device('cuda', 0)
    A = c['board'] * c['board'] + 1
    net = MuZeroBoardGameNet((c['chan'], c['board'], c['board']), A, c['blocks'], c['planes'])
    net.load_state_dict(kinkfree_state_dict(net, 4000 + i))
    net = net.to(dev)
    net.train()
    rs = np.random.RandomState(c['seed'])
    B, K, shape = c['B'], c['K'], (c['chan'], c['board'], c['board'])
    tr = _batch(rs, B, shape, A, K=K, int8_state=c['int8'])
    w = rs.uniform(0.3, 1.0, B).astype(np.float32) if c['weights'] else np.ones(B, np.float32)
    loss_d, prio_d, gd, sd_d, closest = _f64_reference(net, tr._replace(state=tr.state.astype(np.float64)), w, dev)
    err32 = f32_errors(net, tr._replace(state=tr.state.astype(np.float32)), w, dev, gd)
    hl = HipLearner(net, dev, K, B, lr=1e-3)
    loss, prio = hl.grad(_ring(tr, dev), None, torch.from_numpy(w).to(dev) if c['weights'] else None, B)
    assert abs(float(loss) - loss_d) <= 1e-4 * max(1.0, abs(loss_d)), c
    errs = grad_errors(gd, hl.grad_views)
    k, e, bar = kinkfree_worst(errs, err32)  # (TIGHT, or 4 x PyTorch-ROCm's float32 error on an ill-conditioned batch)
    if e > bar:  # only next to a min / max tie of normalize_hidden_state (the weights take care of the ReLUs, not of those)
        assert closest < 2e-6 and e <= 8e-2, (c, k, e, bar, closest)
        _carved('convkf', i, f'{e:.2e} closest {closest:.1e}')
    hl.close()


ATARI_LEARN_CASES_N = int(os.environ.get('MZ_FUZZ_ATARI_LEARN_CASES', '4'))


@pytest.mark.parametrize('i', range(ATARI_LEARN_CASES_N))
def test_random_atari_learner_configuration_matches_float64_autograd(i):
    """Round 5: the conv learner's Atari path (tiles of the 48 x 48 / 24 x 24 stages, parity-plane strided convolutions, pools, categorical heads)
    over frame stacks 1-32, planes 8-128, 1-3 blocks, 3-18 actions, supports 5-601 (value and reward drawn apart), unroll 1-6, batches 1-9, against
    float64 PyTorch-ROCm autograd.  Twice per case (tests/test_gpu_atari_learner.py explains why): kink-free weights against plain float64 autograd
    (every gradient tensor at 2e-3) and seeded random weights against float64 autograd on the HIP pass's own branches (1e-4; round 6)."""
    import torch

    from test_gpu_atari_learner import _case, kinkfree_worst

    rs = np.random.RandomState(8800 + i + 100000 * OFFSET)
    chan, planes, blocks = int(rs.choice([1, 2, 4, 4, 8, 32])), int(rs.choice([8, 16, 24, 40, 64, 128])), int(rs.choice([1, 1, 2, 3]))
    A, vs, rsz, K = int(rs.randint(3, 19)), int(rs.choice([5, 11, 31, 61, 601])), int(rs.choice([5, 11, 31, 61, 601])), int(rs.choice([5, 5, 1, 2, 3, 6]))
    B = int(rs.choice([1, 2, 3, 5, 9]))
    if planes >= 64:
        B = min(B, 3)
    c = (chan, planes, blocks, A, vs, rsz, B, K, int(rs.randint(1 << 20)))
    i8 = bool(rs.rand() < 0.25)  # int8 state storage (drawn after everything else: the earlier cases keep their shapes)
    dev = torch.device('cuda', 0)
    errs, probe = _case(*c, True, dev, i8)
    k, e, bar = kinkfree_worst(errs, probe.err32, probe.closest_tie)  # (TIGHT; 4 x PyTorch-ROCm's float32 error on an ill-conditioned batch; NOISY next to a normalisation tie)
    # (closest_all: the construction's own check -- no ReLU pre-activation anywhere near float32 resolution of zero; seen down to 5e-5 once in 900 cases)
    assert probe.closest_all > 1e-5 and e <= bar, (c, k, e, bar, probe.closest_all, probe.closest_tie)
    from test_gpu_atari_learner import _random_case
    from test_gpu_conv_learner import same_branch_bar

    from test_gpu_atari_learner import ATARI_FLAT

    errs, err32, flipped = _random_case(*c, dev, i8)  # random weights: float64 on the HIP pass's own branches
    wk, worst, bar = same_branch_bar(errs, err32, flat=ATARI_FLAT)
    assert worst <= bar and bar <= 0.1, (c, wk, worst, bar, flipped)  # (bar: 5e-4, or 4 x PyTorch-ROCm float32 on the same branch -- 1.1e-2 on one ill-conditioned case in 700)
