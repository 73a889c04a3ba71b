#!/usr/bin/env python3
"""Learner step (f2) on the classic-control MLP (512 / 64 / 31, unroll 5): ms per update of
  * eager   -- learner.train_step (PyTorch-ROCm autograd + torch.optim.Adam),
  * graphed -- learner.GraphedTrainStep (the same update replayed as one HIP graph),
  * hip     -- hip_learner.HipLearner (hand-written gfx950 kernels, batch gathered from the HBM replay ring by index),
at batch 128 (the reference's classic batch size in this repo's examples) and at large batches, with the HIP step's share of the
fp32 MFMA peak (algorithmic FLOPs: forward 2 MAC-FLOPs, backward twice that, per sample: 3 x 2 x 1 031 168 MAC).
    python tools/learner_bench.py [--batches 128,1024,4096,16384] [--no-torch]"""
import argparse
import copy
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_TF = 157.3


def mac_per_sample(in_dim, A, P, H, Sv, Sr, K):
    rep = in_dim * P + P * H
    step = ((H + A) * P + P * H) + (H * P + P * Sr) + (H * P + P * A) + (H * P + P * Sv)
    return rep + K * step


def timeit(fn, n, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', default='128,1024,4096,16384')
    ap.add_argument('--no-torch', action='store_true')
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--lib', default=None, help='diagnostic A/B: another build of the learner library')
    args = ap.parse_args()
    from muzero_amd import learner
    from muzero_amd.config import make_classic_config
    from muzero_amd import hip_learner
    from muzero_amd.hip_learner import HipLearner
    if args.lib:
        hip_learner.LIB_PATH = os.path.abspath(args.lib)
    from muzero_amd.network import MuZeroMLPNet
    from muzero_amd.replay import Transition

    dev = torch.device('cuda', 0)
    cfg = make_classic_config(use_tensorboard=False)
    K, A = cfg.unroll_steps, 2
    rs = np.random.RandomState(0)
    cap = 50000
    ring = dict(state=torch.from_numpy(rs.uniform(-1, 1, (cap, 20)).astype(np.float32)).to(dev),
                action=torch.from_numpy(rs.randint(0, A, (cap, K)).astype(np.int8)).to(dev),
                pi_prob=torch.from_numpy(rs.dirichlet(np.ones(A), size=(cap, K)).astype(np.float32)).to(dev),
                value=torch.from_numpy(rs.uniform(0, 50, (cap, K)).astype(np.float32)).to(dev),
                reward=torch.ones(cap, K, device=dev))
    flop = 6.0 * mac_per_sample(20, A, cfg.num_planes, cfg.hidden_dim, cfg.value_support_size, cfg.reward_support_size, K)
    out = dict(net='MuZeroMLPNet 512/64/31', unroll=K, flop_per_sample=flop, rows=[])
    for B in [int(b) for b in args.batches.split(',')]:
        torch.manual_seed(0)
        net = MuZeroMLPNet((4, 5), A, cfg.num_planes, cfg.value_support_size, cfg.reward_support_size, cfg.hidden_dim).to(dev)
        row = dict(batch=B)
        idx = torch.from_numpy(rs.randint(0, cap, B).astype(np.int64)).to(dev)
        if not args.no_torch and B <= 4096:
            net_a, net_b = copy.deepcopy(net), copy.deepcopy(net)
            tr = Transition(*[ring[f].index_select(0, idx) for f in Transition._fields])
            tr = tr._replace(state=tr.state.reshape(B, 4, 5))
            w = torch.ones(B, device=dev)
            opt_a = torch.optim.Adam(net_a.parameters(), lr=cfg.lr_init, weight_decay=cfg.weight_decay)
            sch_a = torch.optim.lr_scheduler.MultiStepLR(opt_a, milestones=[10 ** 9], gamma=0.1)
            row['ms_eager'] = timeit(lambda: learner.train_step(cfg, net_a, opt_a, sch_a, dev, tr, w), 50, 10)
            opt_b = learner.make_capturable_adam(net_b, cfg, dev)
            graphed = learner.GraphedTrainStep(cfg, net_b, opt_b, dev, B, (4, 5), K, A)
            row['ms_graphed'] = timeit(lambda: graphed(tr, w), 100, 10)
        hl = HipLearner(net, dev, K, B, lr=cfg.lr_init, weight_decay=cfg.weight_decay, milestones=cfg.lr_milestones, gamma=cfg.lr_decay_rate)
        row['grad_slices'] = int(hl.grads.numel() // hl.total)
        row['ms_hip'] = timeit(lambda: hl.step(ring, idx, None, B, allreduce=False), args.iters)
        row['ms_hip_grad_only'] = timeit(lambda: hl.grad(ring, idx, None, B), args.iters)
        row['samples_per_s_hip'] = B / (row['ms_hip'] * 1e-3)
        row['tflops_hip'] = flop * B / (row['ms_hip'] * 1e-3) / 1e12
        row['mfma_frac_hip'] = row['tflops_hip'] / PEAK_TF
        if 'ms_graphed' in row:
            row['speedup_vs_graphed'] = row['ms_graphed'] / row['ms_hip']
        out['rows'].append(row)
        print(json.dumps(row), flush=True)
        hl.close()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
