"""Is the batch-128 learner step bound by the host's enqueue rate?  Time of the enqueue loop alone vs the loop + drain."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np, torch
from muzero_amd.config import make_classic_config
from muzero_amd.hip_learner import HipLearner
from muzero_amd.network import MuZeroMLPNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device('cuda', 0); cfg = make_classic_config(use_tensorboard=False); K, A = 5, 2
net = MuZeroMLPNet((4, 5), A, 512, 31, 31, 64).to(dev)
rs = np.random.RandomState(0); cap = 5000
ring = dict(state=torch.from_numpy(rs.uniform(-1, 1, (cap, 20)).astype(np.float32)).to(dev), action=torch.from_numpy(rs.randint(0, A, (cap, K)).astype(np.int8)).to(dev),
            pi_prob=torch.from_numpy(rs.dirichlet(np.ones(A), size=(cap, K)).astype(np.float32)).to(dev), value=torch.from_numpy(rs.uniform(0, 50, (cap, K)).astype(np.float32)).to(dev), reward=torch.ones(cap, K, device=dev))
hl = HipLearner(net, dev, K, B, lr=1e-3)
idx = torch.from_numpy(rs.randint(0, cap, B).astype(np.int64)).to(dev)
for _ in range(50): hl.step(ring, idx, None, B, allreduce=False)
torch.cuda.synchronize()
for n in (200, 1000):
    t0 = time.perf_counter()
    for _ in range(n): hl.step(ring, idx, None, B, allreduce=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'B={B} n={n}: enqueue {1e6 * (t1 - t0) / n:.1f} us/step, total {1e6 * (t2 - t0) / n:.1f} us/step, drain after the loop {1e3 * (t2 - t1):.2f} ms')
