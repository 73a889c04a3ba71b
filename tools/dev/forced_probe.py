"""Round 6 probe: HIP conv / Atari learner gradients against float64 autograd that takes the HIP pass's branches (tests/forced_masks.py), next to the
plain float64 comparison.  python tools/dev/forced_probe.py [board|atari|c5|atarifull]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np
import torch

from forced_masks import forced_f64, hip_decisions, tensor_errors
from helpers import seeded_state_dict
from muzero_amd.hip_learner import HipLearner
from muzero_amd.replay import Transition

dev = torch.device('cuda', 0)


def ring(tr):
    B = tr.state.shape[0]
    return dict(state=torch.from_numpy(tr.state).to(dev).reshape(B, -1).contiguous(), action=torch.from_numpy(tr.action).to(dev),
                pi_prob=torch.from_numpy(tr.pi_prob).to(dev), value=torch.from_numpy(tr.value).to(dev), reward=torch.from_numpy(tr.reward).to(dev))


def run(net, tr, w, B, K, tag):
    net = net.to(dev)
    net.train()
    hl = HipLearner(net, dev, K, B, lr=1e-3)
    loss, prio = hl.grad(ring(tr), None, torch.from_numpy(w).to(dev), B)
    t0 = time.time()
    masks, norms = hip_decisions(hl, net, B, K)
    t1 = time.time()
    loss_d, prio_d, gd, flipped = forced_f64(net, tr._replace(state=tr.state.astype(np.float64)), w, dev, masks, norms)
    t2 = time.time()
    errs = tensor_errors(gd, hl.grad_views)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print(f'{tag}: loss hip {float(loss):.8f} f64 {loss_d:.8f}; flipped ReLU decisions {flipped} of {sum(m.size for m in masks)}; read-back {t1 - t0:.1f}s f64 {t2 - t1:.1f}s')
    print('   worst tensors (forced-mask reference):', [(k, f'{e:.2e}') for k, e in worst])
    _, _, g32, _ = forced_f64(net, tr._replace(state=tr.state.astype(np.float32)), w, dev, masks, norms, dtype=torch.float32)
    e32 = tensor_errors(gd, g32)
    print('   PyTorch-ROCm float32 on the same branch, worst:', [(k, f'{e:.2e}') for k, e in sorted(e32.items(), key=lambda kv: -kv[1])[:3]],
          '| HIP / torch-f32 error ratio, worst tensor: %.1f' % max(errs[k] / max(e32[k], 1e-7) for k in errs))
    # plain float64 (its own decisions) for comparison
    import copy
    from muzero_amd import learner
    net_p = copy.deepcopy(net).double()
    net_p.train()
    t = lambda x, dt: torch.from_numpy(np.asarray(x)).to(dev).to(dt)
    lp, _ = learner.loss_tensors(net_p, t(tr.state.astype(np.float64), torch.float64), t(tr.action, torch.int64), t(tr.value, torch.float64), t(tr.reward, torch.float64),
                                 t(tr.pi_prob, torch.float64), t(w, torch.float64))
    lp.backward()
    ep = tensor_errors({k: p.grad for k, p in net_p.named_parameters()}, hl.grad_views)
    print('   worst tensors (plain float64 reference): ', [(k, f'{e:.2e}') for k, e in sorted(ep.items(), key=lambda kv: -kv[1])[:3]])
    hl.close()


which = sys.argv[1] if len(sys.argv) > 1 else 'board'
SEEDS = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0]  # extra seeds for the full-size cases
if which in ('board', 'c5'):
    from muzero_amd.network import MuZeroBoardGameNet
    cases = [(9, 32, 3, 9, 64, 5, 0), (15, 32, 2, 9, 10, 5, 0), (5, 8, 1, 5, 7, 5, 0)] if which == 'board' else [(15, 128, 8, 9, 128, 5, sd) for sd in SEEDS]
    for board, planes, blocks, chan, B, K, sd in cases:
        A = board * board + 1
        net = MuZeroBoardGameNet((chan, board, board), A, blocks, planes)
        net.load_state_dict(seeded_state_dict(net, 300 + board + 1000 * sd))
        rs = np.random.RandomState(board + 1000 * sd)
        tr = Transition(rs.uniform(0, 1, (B, chan, board, board)).astype(np.float32), rs.randint(0, A, (B, K)).astype(np.int16 if A > 128 else np.int8),
                        rs.dirichlet(np.ones(A), size=(B, K)).astype(np.float32), rs.uniform(-1, 1, (B, K)).astype(np.float32), rs.uniform(-1, 1, (B, K)).astype(np.float32))
        w = rs.uniform(0.3, 1.0, B).astype(np.float32)
        run(net, tr, w, B, K, f'board {board} planes {planes} blocks {blocks} batch {B} seed {sd}')
else:
    from muzero_amd.network import MuZeroAtariNet
    cases = [(4, 8, 1, 6, 11, 11, 3, 5, 3), (4, 16, 2, 18, 61, 31, 5, 5, 1), (2, 40, 1, 9, 21, 21, 4, 4, 6)] if which == 'atari' else [(4, 128, 8, 6, 61, 61, 128, 5, 11 + 100 * sd) for sd in SEEDS]
    for chan, planes, blocks, A, vs, rs_, B, K, seed in cases:
        net = MuZeroAtariNet((chan, 96, 96), A, blocks, planes, vs, rs_)
        net.load_state_dict(seeded_state_dict(net, 100 + seed))
        rs = np.random.RandomState(seed)
        tr = Transition(rs.uniform(0, 1, (B, chan, 96, 96)).astype(np.float32), rs.randint(0, A, (B, K)).astype(np.int8),
                        rs.dirichlet(np.ones(A), size=(B, K)).astype(np.float32), (rs.uniform(-1, 1, (B, K)) * 8.0).astype(np.float32), rs.uniform(-1, 1, (B, K)).astype(np.float32))
        w = rs.uniform(0.3, 1.0, B).astype(np.float32)
        run(net, tr, w, B, K, f'atari planes {planes} blocks {blocks} batch {B} seed {seed}')
