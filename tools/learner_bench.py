#!/usr/bin/env python3
"""Learner step (f2): eager `learner.train_step` vs `learner.GraphedTrainStep` (the same update as one HIP graph) on the classic-control
MLP (batch 128, unroll 5): parity of losses / priorities / weights over a few steps from the same start, then ms per step.
    python tools/learner_bench.py"""
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
    from muzero_amd import learner
    from muzero_amd.config import make_classic_config
    from muzero_amd.network import MuZeroMLPNet
    from muzero_amd.replay import Transition

    dev = torch.device('cuda', 0)
    cfg = make_classic_config(use_tensorboard=False)
    B, K, A = 128, cfg.unroll_steps, 2
    torch.manual_seed(0)
    net_a = MuZeroMLPNet((4, 5), A, cfg.num_planes, cfg.value_support_size, cfg.reward_support_size, cfg.hidden_dim).to(dev)
    net_b = copy.deepcopy(net_a)
    rs = np.random.RandomState(0)

    def batch():
        pi = rs.dirichlet(np.ones(A), size=(B, K)).astype(np.float32)
        return Transition(rs.uniform(-1, 1, (B, 4, 5)).astype(np.float32), rs.randint(0, A, (B, K)).astype(np.int8), pi,
                          rs.uniform(0, 50, (B, K)).astype(np.float32), np.ones((B, K), np.float32)), rs.uniform(0.5, 1, B).astype(np.float32)

    opt_a = torch.optim.Adam(net_a.parameters(), lr=cfg.lr_init, weight_decay=cfg.weight_decay)
    sch_a = torch.optim.lr_scheduler.MultiStepLR(opt_a, milestones=[3], gamma=0.1)  # a milestone inside the parity steps
    opt_b = learner.make_capturable_adam(net_b, cfg, dev)
    sch_b = torch.optim.lr_scheduler.MultiStepLR(opt_b, milestones=[3], gamma=0.1)
    graphed = learner.GraphedTrainStep(cfg, net_b, opt_b, dev, B, (4, 5), K, A)
    worst = dict(loss=0.0, prio=0.0, weights=0.0)
    for step in range(6):
        tr, w = batch()
        la, pa = learner.train_step(cfg, net_a, opt_a, sch_a, dev, tr, w)
        lb, pb = graphed(tr, w)
        sch_b.step()
        worst['loss'] = max(worst['loss'], abs(la - float(lb)) / max(1.0, abs(la)))
        worst['prio'] = max(worst['prio'], float(np.abs(pa - pb.cpu().numpy()).max()))
        for (n, x), (_, y) in zip(net_a.state_dict().items(), net_b.state_dict().items()):
            worst['weights'] = max(worst['weights'], float((x - y).abs().max()))
    lr_a, lr_b = sch_a.get_last_lr()[0], float(opt_b.param_groups[0]['lr'])

    def timeit(fn, n=200):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    tr, w = batch()
    trd = Transition(*[torch.as_tensor(x).to(dev) for x in tr])
    wd = torch.as_tensor(w).to(dev)
    ms_eager = timeit(lambda: learner.train_step(cfg, net_a, opt_a, sch_a, dev, trd, wd))
    ms_graph = timeit(lambda: (graphed(trd, wd), sch_b.step()))
    print(json.dumps(dict(parity_worst=worst, lr_after_milestone=[lr_a, lr_b], ms_per_step_eager=ms_eager, ms_per_step_graphed=ms_graph,
                          speedup=ms_eager / ms_graph, batch=B, unroll=K, net='MuZeroMLPNet 512/64/31')))


if __name__ == '__main__':
    main()
