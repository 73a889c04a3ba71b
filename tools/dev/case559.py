import os, sys
os.environ['MZ_FUZZ_SEED_OFFSET'] = '1'
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import copy
import numpy as np, torch
import test_gpu_fuzz as F
from helpers import build_mlp
from muzero_amd import learner
from muzero_amd.hip_learner import HipLearner
from muzero_amd.replay import Transition
c = F._draw_learn_case(559)
print(c)
case, B = c['case'], c['B']; A, K = case[2], 5
dev = torch.device('cuda', 0)
net_a = build_mlp(case).to(dev); net_b = copy.deepcopy(net_a); net_a.train()
rs = np.random.RandomState(c['seed'])
tr = Transition(rs.uniform(-1, 1, (B,) + tuple(case[1])).astype(np.float32), rs.randint(0, A, (B, K)).astype(np.int8),
                rs.dirichlet(np.ones(A), size=(B, K)).astype(np.float32), rs.uniform(-3, 3, (B, K)).astype(np.float32), rs.uniform(-1, 1, (B, K)).astype(np.float32))
w = np.ones(B, np.float32)
la, pa = learner.calc_loss(net_a, dev, tr, torch.from_numpy(w).to(dev)); la.backward()
hl = HipLearner(net_b, dev, K, B, lr=1e-3)
ring = {f: torch.from_numpy(np.ascontiguousarray(getattr(tr, f))).to(dev) for f in Transition._fields}
ring['state'] = ring['state'].reshape(B, -1).contiguous()
lb, pb = hl.grad(ring, None, torch.from_numpy(w).to(dev), B)
for k, p_ in net_a.named_parameters():
    a, b = p_.grad.detach().cpu().numpy().astype(np.float64), hl.grad_views[k].cpu().numpy().astype(np.float64)
    d = np.linalg.norm((a - b).reshape(a.shape[0], -1), axis=1)
    o = np.argsort(d)[::-1][:5]
    print(k, 'norm', np.linalg.norm(a), 'total diff', np.linalg.norm(d), 'worst rows', o.tolist(), d[o].tolist())
# which samples: pre-activations of the representation layer close to zero?
x = torch.from_numpy(tr.state.reshape(B, -1)).to(dev)
z = net_a.represent_net.net[0](x)
zz = z.detach().abs().cpu().numpy()
idx = np.argwhere(zz < 3e-6)
print('near-zero rep pre-activations (sample, unit, |z|):', [(int(i), int(j), float(zz[i, j])) for i, j in idx][:10])
