#!/usr/bin/env python3
"""Re-runs chosen cases of the Atari learner fuzz (tests/test_gpu_fuzz.py) with KINK-FREE weights and prints what the test does not: the batch shape,
the five worst gradient tensors of the HIP step and of PyTorch-ROCm float32 autograd against float64 autograd, and how close the float64 pass comes to
a min / max tie in normalize_hidden_state (the one kink the weights do not remove).
    MZ_FUZZ_SEED_OFFSET=1 python tools/dev/atari_kf_probe.py 7 16 60"""
import copy
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from muzero_amd import learner
    from muzero_amd import network as nw
    from muzero_amd.hip_learner import HipLearner
    from muzero_amd.network import MuZeroAtariNet
    from muzero_amd.replay import Transition
    from test_gpu_atari_learner import _ring, grad_errors, kinkfree_state_dict

    OFFSET = int(os.environ.get('MZ_FUZZ_SEED_OFFSET', '0'))
    dev = torch.device('cuda', 0)
    for i in [int(a) for a in sys.argv[1:]]:
        rs = np.random.RandomState(8800 + i + 100000 * OFFSET)
        chan, planes, blocks = int(rs.choice([1, 2, 4, 4, 8, 32])), int(rs.choice([8, 16, 24, 40, 64, 128])), int(rs.choice([1, 1, 2, 3]))
        A, vs, rsz, K = int(rs.randint(3, 19)), int(rs.choice([5, 11, 31, 61, 601])), int(rs.choice([5, 11, 31, 61, 601])), int(rs.choice([5, 5, 1, 2, 3, 6]))
        B = int(rs.choice([1, 2, 3, 5, 9]))
        if planes >= 64:
            B = min(B, 3)
        seed = int(rs.randint(1 << 20))
        ov = lambda n, v: int(os.environ.get('PROBE_' + n, v))  # noqa: E731  (overrides: PROBE_CHAN / PLANES / BLOCKS / A / VS / RS / B / K / SEED)
        chan, planes, blocks, A, vs, rsz, B, K, seed = ov('CHAN', chan), ov('PLANES', planes), ov('BLOCKS', blocks), ov('A', A), ov('VS', vs), ov('RS', rsz), ov('B', B), ov('K', K), ov('SEED', seed)
        print(f'case {i}: frames {chan} planes {planes} blocks {blocks} actions {A} supports {vs}/{rsz} batch {B} unroll {K} seed {seed}')
        net = MuZeroAtariNet((chan, 96, 96), A, blocks, planes, vs, rsz)
        net.load_state_dict(kinkfree_state_dict(net, 100 + seed))
        net = net.to(dev)
        net.train()
        r2 = np.random.RandomState(seed)
        tr = Transition(r2.uniform(0, 1, (B, chan, 96, 96)).astype(np.float32), r2.randint(0, A, (B, K)).astype(np.int8),
                        r2.dirichlet(np.ones(A), size=(B, K)).astype(np.float32), (r2.uniform(-1, 1, (B, K)) * 8.0).astype(np.float32),
                        r2.uniform(-1, 1, (B, K)).astype(np.float32))
        w = r2.uniform(0.3, 1.0, B).astype(np.float32)
        t = lambda x, dt: torch.from_numpy(np.asarray(x)).to(dev).to(dt)  # noqa: E731
        ties = []
        norm0 = nw.normalize_hidden_state

        def norm_probe(hs):
            v = hs.detach().flatten(2) if hs.dim() > 2 else hs.detach()  # min / max over the channels of each position (util.py:31-36: dim 1)
            top = v.topk(2, dim=1).values
            low = (-v).topk(2, dim=1).values
            ties.append((float((top[:, 0] - top[:, 1]).min()), float((low[:, 0] - low[:, 1]).abs().min()), float((v.max(dim=1).values - v.min(dim=1).values).min())))
            return norm0(hs)

        out = {}
        for name, dt in (('f64', torch.float64), ('f32', torch.float32)):
            n2 = copy.deepcopy(net).to(dt)
            n2.train()
            if name == 'f64':
                nw.normalize_hidden_state = norm_probe
            loss, _ = learner.loss_tensors(n2, t(tr.state, dt), t(tr.action, torch.int64), t(tr.value, dt), t(tr.reward, dt), t(tr.pi_prob, dt), t(w, dt))
            nw.normalize_hidden_state = norm0
            loss.backward()
            out[name] = {k: p.grad for k, p in n2.named_parameters()}
        hl = HipLearner(net, dev, K, B, lr=1e-3)
        hl.grad(_ring(tr, dev), None, torch.from_numpy(w).to(dev), B)
        e_hip = grad_errors(out['f64'], hl.grad_views)
        e_t32 = grad_errors(out['f64'], out['f32'])
        print('   normalisation (max-gap, min-gap, range) per call:', ['%.1e/%.1e/%.1e' % x for x in ties])
        for k in sorted(e_hip, key=e_hip.get, reverse=True)[:int(os.environ.get('PROBE_TOP', '5'))]:
            print(f'   {k:60s} HIP {e_hip[k]:.2e}   torch32 {e_t32[k]:.2e}')
        print('   worst torch32:', max(e_t32.values()))


if __name__ == '__main__':
    main()
