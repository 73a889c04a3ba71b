"""Debug: per-tensor gradient error of the HIP learner vs autograd for one shape."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import copy
import numpy as np, torch
from helpers import build_mlp, mlp_case
from muzero_amd import learner
from muzero_amd.hip_learner import HipLearner
from muzero_amd.replay import Transition
from test_gpu_hip_learner import _random_batch

cname, B, vmax = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
case = mlp_case(cname)
dev = torch.device('cuda', 0)
net_a = build_mlp(case).to(dev); net_b = copy.deepcopy(net_a)
net_a.train()
hl = HipLearner(net_b, dev, 5, B, lr=1e-3)
rs = np.random.RandomState(7)
tr = _random_batch(rs, B, tuple(case[1]), case[2], int8_state=False, vmax=vmax)
w = rs.uniform(0.3, 1.0, B).astype(np.float32)
la0, pa = learner.calc_loss(net_a, dev, tr, torch.from_numpy(w).to(dev))
la0.backward()
ring = dict(state=torch.from_numpy(tr.state).to(dev).reshape(B, -1).contiguous(), action=torch.from_numpy(tr.action).to(dev),
            pi_prob=torch.from_numpy(tr.pi_prob).to(dev), value=torch.from_numpy(tr.value).to(dev), reward=torch.from_numpy(tr.reward).to(dev))
lb0, pb = hl.grad(ring, None, torch.from_numpy(w).to(dev), B)
print('loss', float(la0), float(lb0), 'prio err', np.abs(pa - pb.cpu().numpy()).max())
for k, p in net_a.named_parameters():
    a, b = p.grad.cpu().numpy(), hl.grad_views[k].cpu().numpy()
    d = np.abs(a - b); scale = np.abs(a).max()
    bad = np.argwhere(d > 2e-3 * scale)
    print(f'{k:45s} max|a| {scale:.3e} maxdiff {d.max():.3e} bad {len(bad)}/{a.size}', bad[:6].tolist() if len(bad) else '')
