#!/usr/bin/env python3
"""Learner step of the CONV nets (row f2's PyTorch-ROCm half: the hand-written kernels cover the MLP nets): ms per update of learner.train_step
(eager autograd + Adam) and of learner.GraphedTrainStep (the same update replayed as one HIP graph) for the board-game net.
    python tools/conv_learner_bench.py [--board 15 --planes 128 --blocks 8 --batch 128]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--board', type=int, default=15)
    ap.add_argument('--planes', type=int, default=128)
    ap.add_argument('--blocks', type=int, default=8)
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    from muzero_amd import learner
    from muzero_amd.config import make_gomoku_config
    from muzero_amd.network import MuZeroBoardGameNet
    from muzero_amd.replay import Transition

    dev = torch.device('cuda', 0)
    cfg = make_gomoku_config(use_tensorboard=False)
    N, B, K = args.board, args.batch, cfg.unroll_steps
    A, shape = N * N + 1, (9, N, N)
    torch.manual_seed(0)
    net_a = MuZeroBoardGameNet(shape, A, args.blocks, args.planes).to(dev)
    net_b = copy.deepcopy(net_a)
    rs = np.random.RandomState(0)
    tr = Transition(torch.from_numpy(rs.randint(0, 2, (B,) + shape).astype(np.float32)).to(dev), torch.from_numpy(rs.randint(0, A, (B, K)).astype(np.int16)).to(dev),
                    torch.from_numpy(rs.dirichlet(np.ones(A), size=(B, K)).astype(np.float32)).to(dev), torch.from_numpy(rs.uniform(-1, 1, (B, K)).astype(np.float32)).to(dev),
                    torch.from_numpy(rs.uniform(-1, 1, (B, K)).astype(np.float32)).to(dev))
    w = torch.ones(B, device=dev)

    def timeit(fn, n, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    opt_a = torch.optim.Adam(net_a.parameters(), lr=cfg.lr_init, weight_decay=cfg.weight_decay)
    sch_a = torch.optim.lr_scheduler.MultiStepLR(opt_a, milestones=[10 ** 9], gamma=0.1)
    row = dict(net=f'MuZeroBoardGameNet {N}x{N}, {args.planes} planes, {args.blocks} blocks, A={A}', batch=B, unroll=K,
               parameters=sum(p.numel() for p in net_a.parameters()))
    row['ms_eager'] = timeit(lambda: learner.train_step(cfg, net_a, opt_a, sch_a, dev, tr, w), args.iters)
    opt_b = learner.make_capturable_adam(net_b, cfg, dev)
    graphed = learner.GraphedTrainStep(cfg, net_b, opt_b, dev, B, shape, K, A)
    row['ms_graphed'] = timeit(lambda: graphed(tr, w), args.iters)
    row['samples_per_s_graphed'] = B / (row['ms_graphed'] * 1e-3)
    print(json.dumps(row))


if __name__ == '__main__':
    main()
