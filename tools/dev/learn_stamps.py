"""Phase timeline (shader cycles) of tile 0's workgroups in k_learn_unroll, step k=2: MZL_STAMPS=1 python tools/dev/learn_stamps.py [B]"""
import ctypes as C, os, sys
os.environ['MZL_STAMPS'] = '1'
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np, torch
from muzero_amd.config import make_classic_config
from muzero_amd.hip_learner import HipLearner, load_library
from muzero_amd.network import MuZeroMLPNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device('cuda', 0); cfg = make_classic_config(use_tensorboard=False); K, A = 5, 2
net = MuZeroMLPNet((4, 5), A, 512, 31, 31, 64).to(dev)
rs = np.random.RandomState(0); cap = 5000
ring = dict(state=torch.from_numpy(rs.uniform(-1, 1, (cap, 20)).astype(np.float32)).to(dev), action=torch.from_numpy(rs.randint(0, A, (cap, K)).astype(np.int8)).to(dev),
            pi_prob=torch.from_numpy(rs.dirichlet(np.ones(A), size=(cap, K)).astype(np.float32)).to(dev), value=torch.from_numpy(rs.uniform(0, 50, (cap, K)).astype(np.float32)).to(dev), reward=torch.ones(cap, K, device=dev))
hl = HipLearner(net, dev, K, B, lr=1e-3)
idx = torch.from_numpy(rs.randint(0, cap, B).astype(np.int64)).to(dev)
L = load_library(); L.mzl_debug_stamps.restype = C.c_void_p; L.mzl_debug_stamps.argtypes = [C.c_void_p]
ptr = L.mzl_debug_stamps(hl._h)
for _ in range(20): hl.step(ring, idx, None, B, allreduce=False)
torch.cuda.synchronize()
buf = (C.c_longlong * 64)()
hip = C.CDLL('libamdhip64.so')
hip.hipMemcpy(buf, C.c_void_p(ptr), 512, 2)
st = np.array(buf[:], dtype=np.int64)
# (the last k_learn_unroll launch that stamps is k = K: roles 1, 2 return early there, role 3 runs; role 0 last stamps at k = K-1)
for r, name in enumerate(['dyn', 'policy', 'value', 'reward']):
    v = st[16 * r:16 * r + 8]
    x = st[16 * r + 8:16 * r + 12]
    if x[0] > 0: print('   G1 detail: after ks_load +%d, after save_T +%d, after mma+epi +%d (from barrier 1)' % tuple(int(t - v[1]) for t in x[:3]))
    print(name, 'deltas (cycles):', np.diff(v[v > 0]).tolist(), 'total', int(v[v > 0][-1] - v[0]) if (v > 0).sum() > 1 else None)
