#!/usr/bin/env python3
"""For one case of the Atari learner fuzz with kink-free weights: the worst mean^2 / variance ratio over the channels of every BatchNorm INPUT in the
float64 pass (one-pass variance E[y^2] - mean^2 from float32 partial sums loses log10(ratio) of its 7 digits), next to var itself.
    MZ_FUZZ_SEED_OFFSET=1 python tools/dev/bn_condition_probe.py 16"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from muzero_amd import learner
    from muzero_amd.network import MuZeroAtariNet
    from muzero_amd.replay import Transition
    from test_gpu_atari_learner import kinkfree_state_dict

    OFFSET = int(os.environ.get('MZ_FUZZ_SEED_OFFSET', '0'))
    dev = torch.device('cuda', 0)
    i = int(sys.argv[1])
    rs = np.random.RandomState(8800 + i + 100000 * OFFSET)
    chan, planes, blocks = int(rs.choice([1, 2, 4, 4, 8, 32])), int(rs.choice([8, 16, 24, 40, 64, 128])), int(rs.choice([1, 1, 2, 3]))
    A, vs, rsz, K = int(rs.randint(3, 19)), int(rs.choice([5, 11, 31, 61, 601])), int(rs.choice([5, 11, 31, 61, 601])), int(rs.choice([5, 5, 1, 2, 3, 6]))
    B = int(rs.choice([1, 2, 3, 5, 9]))
    if planes >= 64:
        B = min(B, 3)
    seed = int(rs.randint(1 << 20))
    net = MuZeroAtariNet((chan, 96, 96), A, blocks, planes, vs, rsz)
    net.load_state_dict(kinkfree_state_dict(net, 100 + seed))
    net = net.to(dev).double()
    net.train()
    r2 = np.random.RandomState(seed)
    tr = Transition(r2.uniform(0, 1, (B, chan, 96, 96)), r2.randint(0, A, (B, K)).astype(np.int8), r2.dirichlet(np.ones(A), size=(B, K)),
                    (r2.uniform(-1, 1, (B, K)) * 8.0), r2.uniform(-1, 1, (B, K)))
    w = r2.uniform(0.3, 1.0, B)
    rows = []

    def hook(name):
        def f(_m, inp):
            y = inp[0].detach()
            m = y.mean(dim=(0, 2, 3))
            v = y.var(dim=(0, 2, 3), unbiased=False)
            ratio = (m * m / v.clamp_min(1e-300))
            j = int(ratio.argmax())
            rows.append((float(ratio[j]), name, float(m[j]), float(v[j])))
        return f

    for name, m in net.named_modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.register_forward_pre_hook(hook(name))
    t = lambda x, dt: torch.from_numpy(np.asarray(x)).to(dev).to(dt)  # noqa: E731
    learner.loss_tensors(net, t(tr.state, torch.float64), t(tr.action, torch.int64), t(tr.value, torch.float64), t(tr.reward, torch.float64),
                         t(tr.pi_prob, torch.float64), t(w, torch.float64))
    rows.sort(reverse=True)
    for r in rows[:8]:
        print('mean^2/var %.2e   %-50s mean %.3e var %.3e' % r)


if __name__ == '__main__':
    main()
