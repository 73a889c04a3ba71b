#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP conv learner (muzero_amd/csrc/mz_learn_conv.h) with PyTorch-ROCm autograd on the same batch:
BatchNorm running statistics after the step (one forward check per conv layer), loss, priorities, every gradient tensor.  Prints, no asserts.
    python tools/dev/conv_learner_check.py [--board 5 --planes 8 --blocks 1 --batch 7 --chan 5]"""
import argparse
import copy
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--board', type=int, default=5)
    ap.add_argument('--planes', type=int, default=8)
    ap.add_argument('--blocks', type=int, default=1)
    ap.add_argument('--batch', type=int, default=7)
    ap.add_argument('--chan', type=int, default=5)
    ap.add_argument('--unroll', type=int, default=5)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--int8', action='store_true')
    ap.add_argument('--atari', action='store_true', help='MuZeroAtariNet on 96 x 96 frames (--chan frames, --actions, --support)')
    ap.add_argument('--actions', type=int, default=6)
    ap.add_argument('--support', type=int, default=11)
    ap.add_argument('--brief', action='store_true', help='one line per network (representation / dynamics / prediction) instead of one per tensor')
    args = ap.parse_args()
    from helpers import seeded_state_dict
    from muzero_amd import learner
    from muzero_amd.hip_learner import HipLearner
    from muzero_amd.network import MuZeroAtariNet, MuZeroBoardGameNet

    dev = torch.device('cuda', 0)
    N, B, K = args.board, args.batch, args.unroll
    A, shape = N * N + 1, (args.chan, N, N)
    if args.atari:
        A, shape = args.actions, (args.chan, 96, 96)
        net_a = MuZeroAtariNet(shape, A, args.blocks, args.planes, args.support, args.support)
    else:
        net_a = MuZeroBoardGameNet(shape, A, args.blocks, args.planes)
    net_a.load_state_dict(seeded_state_dict(net_a, 100 + args.seed))
    net_a = net_a.to(dev)
    net_b = copy.deepcopy(net_a)
    net_a.train()
    net_b.train()
    rs = np.random.RandomState(args.seed)
    st = rs.randint(0, 2, (B,) + shape).astype(np.int8) if args.int8 else rs.uniform(0, 1, (B,) + shape).astype(np.float32)
    ac = rs.randint(0, A, (B, K)).astype(np.int16 if A > 128 else np.int8)
    pi = rs.dirichlet(np.ones(A), size=(B, K)).astype(np.float32)
    va = rs.uniform(-1, 1, (B, K)).astype(np.float32) * (8.0 if args.atari else 1.0)
    re = rs.uniform(-1, 1, (B, K)).astype(np.float32)
    w = rs.uniform(0.3, 1.0, B).astype(np.float32)
    t = lambda x: torch.from_numpy(x).to(dev)  # noqa: E731
    loss_a, prio_a = learner.loss_tensors(net_a, t(st.astype(np.float32)), t(ac.astype(np.int64)), t(va), t(re), t(pi), t(w))
    loss_a.backward()
    # float64 autograd on the same batch: separates rounding / ReLU-kink noise (torch fp32 vs fp64) from bugs (HIP vs fp64)
    net_d = copy.deepcopy(net_b).double()
    net_d.train()
    import torch.nn.functional as F
    relu0, near = F.relu, []

    def relu_probe(x, inplace=False):  # how close the float64 pass comes to a ReLU kink, per call
        nz = x.detach().abs()
        nz = nz[nz > 0]
        near.append((float(nz.min()) if nz.numel() else float('inf'), int((nz < 1e-5).sum()), tuple(x.shape)))
        return relu0(x, inplace=False)

    F.relu = relu_probe
    loss_d, prio_d = learner.loss_tensors(net_d, t(st.astype(np.float64)), t(ac.astype(np.int64)), t(va).double(), t(re).double(), t(pi).double(), t(w).double())
    F.relu = relu0
    loss_d.backward()
    near.sort()
    print('closest ReLU pre-activations (float64 pass): ' + '  '.join(f'{m:.1e} (x{c} < 1e-5, shape {list(sh)})' for m, c, sh in near[:3]))
    gd = {k: p.grad for k, p in net_d.named_parameters()}
    hl = HipLearner(net_b, dev, K, B, lr=1e-3)
    ring = dict(state=t(st).reshape(B, -1).contiguous(), action=t(ac), pi_prob=t(pi), value=t(va), reward=t(re))
    loss_b, prio_b = hl.grad(ring, None, t(w), B)
    torch.cuda.synchronize()
    print('loss', float(loss_a), float(loss_b))
    print('prio max abs diff', float((prio_a - prio_b).abs().max()), 'scale', float(prio_a.abs().max()))
    sd_a, sd_b = net_a.state_dict(), net_b.state_dict()
    if not args.brief:
        print('---- running statistics (forward, per layer) ----')
    stat = {}
    for k in sd_a:
        if 'running' in k or 'num_batches' in k:
            a, b = sd_a[k].float(), sd_b[k].float()
            e = float((a - b).abs().max()) / max(float(a.abs().max()), 1e-12)
            stat[k.split('.')[0]] = max(stat.get(k.split('.')[0], 0.0), e)
            if not args.brief:
                print(f'{k:64s} err {float((a - b).abs().max()):.3e}  scale {float(a.abs().max()):.3e}')
    if not args.brief:
        print('---- gradients ----')
    worst = worst_t = 0.0
    grp = {}
    for k, p in net_a.named_parameters():
        a, b, d = p.grad, hl.grad_views[k], gd[k]
        sc = float(d.abs().max())
        err, err_t = float((d - b.double()).abs().max()), float((d - a.double()).abs().max())
        worst, worst_t = max(worst, err / max(sc, 1e-12)), max(worst_t, err_t / max(sc, 1e-12))
        g = grp.setdefault(k.split('.')[0], [0.0, 0.0])
        g[0], g[1] = max(g[0], err / max(sc, 1e-12)), max(g[1], err_t / max(sc, 1e-12))
        if not args.brief:
            print(f'{k:64s} scale {sc:.3e}  HIP-vs-f64 {err / max(sc, 1e-12):.2e}  torch32-vs-f64 {err_t / max(sc, 1e-12):.2e}')
    for k, g in grp.items():
        print(f'{k:16s} running-stat rel err {stat.get(k, 0.0):.2e}   grad HIP-vs-f64 {g[0]:.2e}   torch32-vs-f64 {g[1]:.2e}')
    print('worst relative gradient error: HIP', worst, ' torch fp32', worst_t)


if __name__ == '__main__':
    main()
