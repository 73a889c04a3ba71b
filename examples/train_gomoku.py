#!/usr/bin/env python3
"""End-to-end MuZero on Gomoku 9x9 (five in a row) with a small conv (board) network: the conv-tower planner path -- MFMA conv
kernels / fused residual tower, HBM-resident trees, sparse action planes, device Gomoku env -- driving the same single-process
loop as examples/train_tictactoe.py.  The reference's gomoku hyper-parameters (config.py:139-167) with a smaller network
(32 planes, 2 blocks) and 32 simulations so that a few minutes show learning.  Evaluation: deterministic games against a uniformly
random opponent as black and as white (win = five in a row before the opponent; the random player never resigns).

By default the trajectories never leave the GPU: the device epilogue writes (Transition, priority) items into an HBM replay ring
(`Planner.attach_replay`; int16 action fields where num_actions > 128, e.g. --board 15), the batch indices are drawn on the device
(`replay.DeviceSampler`) and the update runs on the hand-written conv-learner kernels (round 5: `hip_learner.HipLearner`, csrc/mz_learn_conv.h:
train-mode BatchNorm towers, heads, losses, backward, Adam -- no ATen kernel on the step).  --autograd keeps the PyTorch-ROCm update
(learner.train_step), --graphed replays it as ONE HIP graph (learner.GraphedTrainStep, captured before the loop).  --host-assembly keeps the
reference's host-side assembler (pipeline.py:118-165).

    python examples/train_gomoku.py --train-steps 2000 --envs 128 [--autograd | --graphed] [--board 9]"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def play_vs_random(net, dev, cfg, agent_player, games, rs, board=9):
    from muzero_amd import mcts
    from muzero_amd.games import GomokuEnv

    res = dict(win=0, draw=0, loss=0)
    for _ in range(games):
        env = GomokuEnv(board_size=board)
        obs, done = env.reset(), False
        while not done:
            if env.current_player == agent_player:
                action, *_ = mcts.uct_search(obs, net, dev, cfg, 0.0, env.actions_mask, env.current_player, env.opponent_player, deterministic=True)
            else:
                legal = np.flatnonzero(env.actions_mask[:board * board])  # the random opponent never resigns
                action = int(rs.choice(legal))
            obs, _, done, _ = env.step(action)
        res['draw' if env.winner is None else ('win' if env.winner == agent_player else 'loss')] += 1
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--train-steps', type=int, default=4000)
    ap.add_argument('--envs', type=int, default=128)
    ap.add_argument('--moves-per-iter', type=int, default=8)
    ap.add_argument('--updates-per-iter', type=int, default=16)
    ap.add_argument('--report-every', type=int, default=500)
    ap.add_argument('--eval-games', type=int, default=10)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--board', type=int, default=9)
    ap.add_argument('--host-assembly', action='store_true', help='trajectories through selfplay_read + the host EpisodeAssembler instead of the device epilogue')
    ap.add_argument('--graphed', action='store_true', help='the PyTorch-ROCm update as one HIP graph (learner.GraphedTrainStep)')
    ap.add_argument('--autograd', action='store_true', help='the PyTorch-ROCm update, eager (learner.train_step)')
    ap.add_argument('--out', default='')
    args = ap.parse_args()

    from muzero_amd import learner
    from muzero_amd import planner as pl
    from muzero_amd.config import make_gomoku_config
    from muzero_amd.network import MuZeroBoardGameNet
    from muzero_amd.pipeline import EpisodeAssembler
    from muzero_amd.replay import PrioritizedReplay

    torch.manual_seed(args.seed)
    dev = torch.device('cuda', 0)
    cfg = make_gomoku_config(num_training_steps=args.train_steps, batch_size=128, min_replay_size=5000, use_tensorboard=False)
    cfg.num_envs, cfg.num_simulations, cfg.num_planes, cfg.num_res_blocks = args.envs, 32, 32, 2
    N = args.board
    A, obs_shape = N * N + 1, (9, N, N)
    net = MuZeroBoardGameNet(obs_shape, A, cfg.num_res_blocks, cfg.num_planes).to(dev)
    graphed = hip = None
    use_hip = not (args.graphed or args.autograd or args.host_assembly)
    if args.graphed:  # captured here, before anything else drives the GPU (learner.prepare_graphed_step)
        import warnings

        warnings.filterwarnings('ignore', message='Detected call of `lr_scheduler.step\\(\\)` before')  # (optimizer.step() runs inside the graph)
        opt = learner.make_capturable_adam(net, cfg, dev)
        graphed = learner.prepare_graphed_step(cfg, net, opt, dev, obs_shape, A)
    elif use_hip:  # the conv learner's kernels; the module's parameters and BatchNorm buffers become views of the learner's flat vectors
        hip = learner.make_hip_learner(cfg, net, dev)
        opt = None
    else:
        opt = torch.optim.Adam(net.parameters(), lr=cfg.lr_init, weight_decay=cfg.weight_decay)
    sched = None if hip is not None else torch.optim.lr_scheduler.MultiStepLR(opt, milestones=cfg.lr_milestones, gamma=cfg.lr_decay_rate)
    replay = PrioritizedReplay(20000, 0.0, 0.0, np.random.RandomState(args.seed), device='cuda')
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), cfg, num_envs=args.envs, seed=args.seed), 0)
    net.eval()
    p.load_state_dict(net.state_dict())
    if not args.host_assembly:
        p.attach_replay(replay, cfg, obs_shape=obs_shape)
    p.selfplay_reset(pl.ENV_GOMOKU)
    sampler = replay.device_sampler(seed=args.seed) if hip is not None else None
    asm = EpisodeAssembler(cfg, args.envs, obs_shape)
    rs = np.random.RandomState(args.seed + 7)

    def evaluate(player):
        return play_vs_random(net, dev, cfg, player, args.eval_games, rs, N)

    log = [dict(train_steps=0, black=evaluate(1), white=evaluate(2))]
    print(json.dumps(log[0]), flush=True)
    steps, t0 = 0, time.time()
    while steps < args.train_steps:
        p.selfplay_step(-1.0, args.moves_per_iter)  # per-env temperature schedule of the game (config.py:244-249)
        if args.host_assembly:
            for tr, prio in asm.feed(p.selfplay_read(args.moves_per_iter)):
                replay.add(tr, prio)
        else:
            p.synchronize()  # event order: the moves above are committed to the ring before a batch is drawn
        if replay.size < cfg.min_replay_size:
            continue
        net.train()
        for _ in range(args.updates_per_iter):
            if hip is not None:  # draw, gather, update: all on the device, nothing read back
                idx_t, w_t, ring = sampler.sample(cfg.batch_size)
                loss, prio = hip.step(ring, idx_t, w_t, cfg.batch_size)
                steps += 1
                if steps % args.report_every == 0:
                    net.eval()
                    rec = dict(train_steps=steps, loss=float(loss), seconds=round(time.time() - t0, 1), env_steps=p.selfplay_counters()['env_steps'],
                               black=evaluate(1), white=evaluate(2), learner='hip')
                    log.append(rec)
                    print(json.dumps(rec), flush=True)
                continue
            batch, idx, w = replay.sample_tensors(cfg.batch_size)
            if graphed is not None:  # uniform replay (the launchers' default) ignores priorities: no host read-back per update
                loss, prio = graphed(batch, w)
                sched.step()
            else:
                loss, prio = learner.train_step(cfg, net, opt, sched, dev, batch, w)
                replay.update_priorities(idx, prio)
            steps += 1
            if steps % args.report_every == 0:
                net.eval()
                rec = dict(train_steps=steps, loss=float(loss), seconds=round(time.time() - t0, 1), env_steps=p.selfplay_counters()['env_steps'],
                           black=evaluate(1), white=evaluate(2))
                net.train()
                log.append(rec)
                print(json.dumps(rec), flush=True)
        net.eval()
        p.load_state_dict(net.state_dict())
    if args.out:
        json.dump(dict(args=vars(args), log=log), open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
