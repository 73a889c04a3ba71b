#!/usr/bin/env python3
"""Re-runs chosen cases of tests/test_gpu_fuzz.py's Atari learner fuzz with several data seeds and prints, per network, the HIP step's and PyTorch-ROCm
float32 autograd's gradient error against float64 autograd: a configuration-dependent bug fails every data seed, mask noise only some.
    python tools/dev/atari_fuzz_probe.py 0 6 9"""
import copy
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from helpers import seeded_state_dict
    from muzero_amd import learner
    from muzero_amd.hip_learner import HipLearner
    from muzero_amd.network import MuZeroAtariNet
    from muzero_amd.replay import Transition
    from test_gpu_atari_learner import _f64, _ring

    dev = torch.device('cuda', 0)
    for i in [int(a) for a in sys.argv[1:]]:
        rs = np.random.RandomState(8800 + i)
        chan, planes, blocks = int(rs.choice([1, 2, 4, 4, 8, 32])), int(rs.choice([8, 16, 24, 40, 64, 128])), int(rs.choice([1, 1, 2, 3]))
        A, vs, rsz, K = int(rs.randint(3, 19)), int(rs.choice([5, 11, 31, 61, 601])), int(rs.choice([5, 11, 31, 61, 601])), int(rs.choice([5, 5, 1, 2, 3, 6]))
        B = int(rs.choice([1, 2, 3, 5, 9]))
        if planes >= 64:
            B = min(B, 3)
        print(f'case {i}: frames {chan} planes {planes} blocks {blocks} actions {A} supports {vs}/{rsz} unroll {K} batch {B}')
        for ds in range(5):
            net = MuZeroAtariNet((chan, 96, 96), A, blocks, planes, vs, rsz)
            net.load_state_dict(seeded_state_dict(net, 5000 + i))
            net = net.to(dev)
            net.train()
            r2 = rs if ds == 0 else np.random.RandomState(31337 * ds + i)
            scale = float((vs - 1) // 2) * 0.8
            tr = Transition(r2.uniform(0, 1, (B, chan, 96, 96)).astype(np.float32), r2.randint(0, A, (B, K)).astype(np.int8), r2.dirichlet(np.ones(A), size=(B, K)).astype(np.float32),
                            (r2.uniform(-1, 1, (B, K)) * scale).astype(np.float32), r2.uniform(-1, 1, (B, K)).astype(np.float32))
            w = r2.uniform(0.3, 1.0, B).astype(np.float32)
            loss_d, prio_d, gd, sd_d, probe = _f64(net, tr._replace(state=tr.state.astype(np.float64)), w, dev)
            net_t = copy.deepcopy(net)
            t = lambda x, dt: torch.from_numpy(np.asarray(x)).to(dev).to(dt)  # noqa: E731
            lt, _ = learner.loss_tensors(net_t, t(tr.state, torch.float32), t(tr.action, torch.int64), t(tr.value, torch.float32), t(tr.reward, torch.float32),
                                         t(tr.pi_prob, torch.float32), t(w, torch.float32))
            lt.backward()
            gt = {k: p.grad for k, p in net_t.named_parameters()}
            hl = HipLearner(net, dev, K, B, lr=1e-3)
            hl.grad(_ring(tr, dev), None, torch.from_numpy(w).to(dev), B)
            grp = {}
            for k, g in gd.items():
                sc = max(1e-8, float(g.abs().max()))
                e = grp.setdefault(k.split('.')[0][:4], [0.0, 0.0])
                e[0] = max(e[0], float((g - hl.grad_views[k].double()).abs().max()) / sc)
                e[1] = max(e[1], float((g - gt[k].double()).abs().max()) / sc)
            print(f'   data {ds}: closest all {probe.closest_all:.1e} small {probe.closest_small:.1e}   ' +
                  '   '.join(f'{k}: HIP {e[0]:.1e} torch32 {e[1]:.1e}' for k, e in grp.items()))


if __name__ == '__main__':
    main()
