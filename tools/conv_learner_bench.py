#!/usr/bin/env python3
"""Learner step of the board-game CONV net: ms per update of the HIP learner (muzero_amd/csrc/mz_learn_conv.h, row f2) with its fraction of the
fp32 MFMA peak, beside learner.train_step (eager PyTorch-ROCm autograd + Adam: MIOpen) and learner.GraphedTrainStep (the same update as one HIP graph).
    python tools/conv_learner_bench.py [--board 15 --planes 128 --blocks 8 --batch 128] [--hip-only]"""
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
    ap.add_argument('--hip-only', action='store_true')
    ap.add_argument('--chan', type=int, default=9)
    ap.add_argument('--atari', action='store_true', help='MuZeroAtariNet on 96 x 96 frames (--chan frames, --actions, --support; unroll from the Atari config)')
    ap.add_argument('--actions', type=int, default=6)
    ap.add_argument('--support', type=int, default=61)
    args = ap.parse_args()
    from muzero_amd import learner
    from muzero_amd.config import make_atari_config, make_gomoku_config
    from muzero_amd.network import MuZeroAtariNet, MuZeroBoardGameNet
    from muzero_amd.replay import Transition

    dev = torch.device('cuda', 0)
    cfg = (make_atari_config if args.atari else make_gomoku_config)(use_tensorboard=False)
    N, B, K = args.board, args.batch, cfg.unroll_steps
    A, shape = N * N + 1, (args.chan, N, N)
    torch.manual_seed(0)
    if args.atari:
        A, shape = args.actions, (args.chan, 96, 96)
        net_a = MuZeroAtariNet(shape, A, args.blocks, args.planes, args.support, args.support).to(dev)
        name = f'MuZeroAtariNet {args.chan}x96x96, {args.planes} planes, {args.blocks} blocks, A={A}, supports {args.support}'
    else:
        net_a = MuZeroBoardGameNet(shape, A, args.blocks, args.planes).to(dev)
        name = f'MuZeroBoardGameNet {N}x{N}, {args.planes} planes, {args.blocks} blocks, A={A}'
    net_b = copy.deepcopy(net_a)
    rs = np.random.RandomState(0)
    tr = Transition(torch.from_numpy((rs.uniform(0, 1, (B,) + shape) if args.atari else rs.randint(0, 2, (B,) + shape)).astype(np.float32)).to(dev), torch.from_numpy(rs.randint(0, A, (B, K)).astype(np.int16)).to(dev),
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

    from muzero_amd import hip_learner as _hl
    if os.environ.get('MZL_LIB_PATH'):  # A/B runs: another build of the learner library (tools/dev/halo_ab.sh)
        _hl.LIB_PATH = os.environ['MZL_LIB_PATH']
    from muzero_amd.hip_learner import HipLearner, atari_learner_flops, conv_learner_flops

    net_h = copy.deepcopy(net_a)
    hl = HipLearner(net_h, dev, K, B, lr=cfg.lr_init, weight_decay=cfg.weight_decay)
    ring = dict(state=(tr.state if args.atari else tr.state.to(torch.int8)).reshape(B, -1).contiguous(), action=tr.action, pi_prob=tr.pi_prob, value=tr.value, reward=tr.reward)

    def hip_step():
        hl.grad(ring, None, w, B)
        hl.apply()

    flops = (atari_learner_flops if args.atari else conv_learner_flops)(shape, A, args.blocks, args.planes, K) * B
    ms = timeit(hip_step, args.iters)
    hip = dict(ms_hip=ms, samples_per_s_hip=B / (ms * 1e-3), flop_per_update=flops, tflops=flops / (ms * 1e-3) / 1e12, mfma_frac=flops / (ms * 1e-3) / 157.3e12)
    if args.hip_only:
        print(json.dumps(dict(net=name, batch=B, unroll=K, **hip)))
        return
    opt_a = torch.optim.Adam(net_a.parameters(), lr=cfg.lr_init, weight_decay=cfg.weight_decay)
    sch_a = torch.optim.lr_scheduler.MultiStepLR(opt_a, milestones=[10 ** 9], gamma=0.1)
    row = dict(net=name, batch=B, unroll=K,
               parameters=sum(p.numel() for p in net_a.parameters()))
    row['ms_eager'] = timeit(lambda: learner.train_step(cfg, net_a, opt_a, sch_a, dev, tr, w), args.iters)
    opt_b = learner.make_capturable_adam(net_b, cfg, dev)
    graphed = learner.GraphedTrainStep(cfg, net_b, opt_b, dev, B, shape, K, A)
    row['ms_graphed'] = timeit(lambda: graphed(tr, w), args.iters)
    row['samples_per_s_graphed'] = B / (row['ms_graphed'] * 1e-3)
    row.update(hip)
    print(json.dumps(row))


if __name__ == '__main__':
    main()
