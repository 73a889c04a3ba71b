#!/usr/bin/env python3
"""End-to-end MuZero on TicTacToe (two-player path: player switching, sign-flipping backup, Monte-Carlo returns, bounds
(-1, 1)) with the MI355X planner, same single-process loop as examples/train_cartpole.py.  Hyper-parameters: the reference's
tictactoe config (config.py:106-136, tictactoe/run_training.py:34-59).  Evaluation: deterministic games against a uniformly
random opponent, as black and as white; an agent that has learnt the game almost never loses.

    python examples/train_tictactoe.py --train-steps 4000 --envs 256"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def play_vs_random(net, dev, cfg, agent_player, games, rs):
    from muzero_amd import mcts
    from muzero_amd.games import TicTacToeEnv

    res = dict(win=0, draw=0, loss=0)
    for _ in range(games):
        env = TicTacToeEnv()
        obs, done = env.reset(), False
        while not done:
            if env.current_player == agent_player:
                action, *_ = mcts.uct_search(obs, net, dev, cfg, 0.0, env.actions_mask, env.current_player, env.opponent_player, deterministic=True)
            else:
                legal = np.flatnonzero(env.actions_mask[:9])  # the random opponent never resigns
                action = int(rs.choice(legal))
            obs, _, done, _ = env.step(action)
        res['draw' if env.winner is None else ('win' if env.winner == agent_player else 'loss')] += 1
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--train-steps', type=int, default=4000)
    ap.add_argument('--envs', type=int, default=256)
    ap.add_argument('--moves-per-iter', type=int, default=8)
    ap.add_argument('--updates-per-iter', type=int, default=16)
    ap.add_argument('--report-every', type=int, default=500)
    ap.add_argument('--eval-games', type=int, default=50)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--out', default='')
    ap.add_argument('--host-assembly', action='store_true',
                    help="round 1's path: (Transition, priority) items assembled on the host and the PyTorch learner step; default: device epilogue "
                         'into the HBM replay + the HIP learner kernels (hip_learner.HipLearner), in event order (a seed gives one run)')
    args = ap.parse_args()

    from muzero_amd import learner
    from muzero_amd import planner as pl
    from muzero_amd.config import make_tictactoe_config
    from muzero_amd.network import MuZeroMLPNet
    from muzero_amd.pipeline import EpisodeAssembler
    from muzero_amd.replay import PrioritizedReplay

    torch.manual_seed(args.seed)
    dev = torch.device('cuda', 0)
    cfg = make_tictactoe_config(num_training_steps=args.train_steps, batch_size=128, min_replay_size=5000, use_tensorboard=False)
    cfg.num_envs = args.envs
    net = MuZeroMLPNet((9, 3, 3), 10, cfg.num_planes, cfg.value_support_size, cfg.reward_support_size, cfg.hidden_dim).to(dev)
    hl = None
    if args.host_assembly:
        opt = torch.optim.Adam(net.parameters(), lr=cfg.lr_init, weight_decay=cfg.weight_decay)
        sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=cfg.lr_milestones, gamma=cfg.lr_decay_rate)
    else:
        hl = learner.make_hip_learner(cfg, net, dev)
    replay = PrioritizedReplay(20000, 0.0, 0.0, np.random.RandomState(args.seed), device='cuda')
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), cfg, num_envs=args.envs, seed=args.seed), 0)
    net.eval()
    p.load_state_dict(net.state_dict())
    if hl is not None:
        p.attach_replay(replay, cfg, obs_shape=(9, 3, 3))
    p.selfplay_reset(pl.ENV_TICTACTOE)
    asm = EpisodeAssembler(cfg, args.envs, (9, 3, 3))
    rs = np.random.RandomState(args.seed + 7)

    log = [dict(train_steps=0, black=play_vs_random(net, dev, cfg, 1, args.eval_games, rs), white=play_vs_random(net, dev, cfg, 2, args.eval_games, rs))]
    print(json.dumps(log[0]), flush=True)
    steps, t0 = 0, time.time()
    while steps < args.train_steps:
        p.selfplay_step(-1.0, args.moves_per_iter)  # per-env temperature schedule of the game (config.py:236-241)
        if hl is None:
            for tr, prio in asm.feed(p.selfplay_read(args.moves_per_iter)):
                replay.add(tr, prio)
        else:
            p.synchronize()  # event order: the moves above are committed before a batch is drawn
        if replay.size < cfg.min_replay_size:
            continue
        net.train()
        for _ in range(args.updates_per_iter):
            if hl is not None:
                idx, _, ring = replay.sample_indices(cfg.batch_size)
                loss, prio = hl.step(ring, torch.from_numpy(idx).to(dev), None, cfg.batch_size)
            else:
                batch, idx, w = replay.sample_tensors(cfg.batch_size)
                loss, prio = learner.train_step(cfg, net, opt, sched, dev, batch, w)
                replay.update_priorities(idx, prio)
            steps += 1
            if steps % args.report_every == 0:
                net.eval()
                rec = dict(train_steps=steps, loss=float(loss), seconds=round(time.time() - t0, 1), env_steps=p.selfplay_counters()['env_steps'],
                           black=play_vs_random(net, dev, cfg, 1, args.eval_games, rs), white=play_vs_random(net, dev, cfg, 2, args.eval_games, rs))
                net.train()
                log.append(rec)
                print(json.dumps(rec), flush=True)
        net.eval()
        p.load_state_dict(net.state_dict())
    if args.out:
        json.dump(dict(args=vars(args), log=log), open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
