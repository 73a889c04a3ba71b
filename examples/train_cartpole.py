#!/usr/bin/env python3
"""End-to-end MuZero on CartPole-v1 with the MI355X planner: device-resident self-play (search + env + records in HBM) ->
device epilogue (the reference's n-step targets / priorities / unroll sequences built on the GPU, written straight into the
HBM-resident replay; `--host-assembly` uses the host EpisodeAssembler instead) -> learner step (the reference's loss, Adam,
MultiStepLR) -> weights back into the planner.  One process, one GPU, no actor processes:
the roles of classic/run_training.py's six actors, data collector thread and learner thread are interleaved in one loop.

    python examples/train_cartpole.py --train-steps 3000 --envs 128

Hyper-parameters are the reference's classic config (config.py:170-201, classic/run_training.py:33-56): 50 simulations,
discount 0.997, td_steps 10, unroll 5, Adam lr 5e-3 / weight decay 1e-4, batch 128, replay 50 000, uniform sampling,
temperature 1.0 for the first 30 k training steps.  Prints one JSON line per report with the mean length of the episodes
that finished since the previous report (500 is the environment's cap)."""
import argparse
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
    ap.add_argument('--train-steps', type=int, default=3000)
    ap.add_argument('--envs', type=int, default=128)
    ap.add_argument('--moves-per-iter', type=int, default=8)
    ap.add_argument('--updates-per-iter', type=int, default=16)
    ap.add_argument('--report-every', type=int, default=250)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--eval-episodes', type=int, default=3)
    ap.add_argument('--eval-every', type=int, default=0, help='also evaluate every N training steps (deterministic episodes on the host env); 0: only at the end')
    ap.add_argument('--out', default='')
    ap.add_argument('--host-assembly', action='store_true', help='assemble (Transition, priority) items on the host instead of the device epilogue')
    ap.add_argument('--learner', choices=('hip', 'graphed', 'eager'), default='hip',
                    help='hip: hand-written gfx950 kernels (hip_learner.HipLearner, batch gathered from the HBM ring by index); graphed: the PyTorch '
                         'update as one HIP graph (learner.GraphedTrainStep); eager: learner.train_step')
    ap.add_argument('--overlap', action='store_true',
                    help='let self-play (planner stream) and updates (learner stream) overlap: faster, but how many items the replay holds when a '
                         'batch is drawn then depends on timing -- without it every draw happens after the moves before it have been committed '
                         '(event order), and a seed gives ONE learning curve')
    args = ap.parse_args()
    args.eager_learner = args.learner == 'eager'

    from muzero_amd import learner
    from muzero_amd import planner as pl
    from muzero_amd.config import make_classic_config
    from muzero_amd.network import MuZeroMLPNet
    from muzero_amd.pipeline import EpisodeAssembler
    from muzero_amd.replay import PrioritizedReplay

    torch.manual_seed(args.seed)
    dev = torch.device('cuda', 0)
    cfg = make_classic_config(num_training_steps=args.train_steps, batch_size=128, min_replay_size=5000, use_tensorboard=False)
    cfg.num_envs = args.envs
    net = MuZeroMLPNet((4, 5), 2, cfg.num_planes, cfg.value_support_size, cfg.reward_support_size, cfg.hidden_dim).to(dev)
    hl = None
    if args.learner == 'hip':
        hl = learner.make_hip_learner(cfg, net, dev)
        opt, sched = hl.optimizer, hl.lr_scheduler
    else:
        opt = (torch.optim.Adam(net.parameters(), lr=cfg.lr_init, weight_decay=cfg.weight_decay) if args.eager_learner
               else learner.make_capturable_adam(net, cfg, dev))
        sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=cfg.lr_milestones, gamma=cfg.lr_decay_rate)
    replay = PrioritizedReplay(50000, 0.0, 0.0, np.random.RandomState(args.seed), device='cuda')

    p = pl.Planner(pl.make_mz_config(net.planner_spec(), cfg, num_envs=args.envs, seed=args.seed), 0)
    net.eval()
    p.load_state_dict(net.state_dict())
    if not args.host_assembly:
        p.attach_replay(replay, cfg, obs_shape=(4, 5))
    p.selfplay_reset(pl.ENV_CARTPOLE)
    asm = EpisodeAssembler(cfg, args.envs, (4, 5))

    from muzero_amd import mcts
    from muzero_amd.games import CartPoleEnv

    def evaluate():
        # deterministic evaluation episodes (pipeline.py:400-488: argmax of the visit counts, no root noise) on the host env
        net.eval()
        lengths = []
        for ep in range(args.eval_episodes):
            env = CartPoleEnv(4, seed=1000 + ep)
            obs, done = env.reset(), False
            while not done:
                action, *_ = mcts.uct_search(obs, net, dev, cfg, 0.0, env.actions_mask, 1, 1, deterministic=True)
                obs, _, done, _ = env.step(action)
            lengths.append(env.steps)
        return lengths

    evals = []
    steps, t0, last = 0, time.time(), dict(episodes=0, episode_steps=0)
    graphed = None
    log = []
    while steps < args.train_steps:
        T = float(cfg.visit_softmax_temperature_fn(0, steps))
        p.selfplay_step(T, args.moves_per_iter)
        if args.host_assembly:
            for tr, prio in asm.feed(p.selfplay_read(args.moves_per_iter)):
                replay.add(tr, prio)
        if not args.overlap:
            p.synchronize()  # event order: the moves above are committed to the replay before anything is drawn from it
        if replay.size < cfg.min_replay_size:
            continue
        net.train()
        for _ in range(args.updates_per_iter):
            if hl is not None:  # indices drawn on the host (the reference's RandomState stream), items gathered by the kernels
                idx, _, ring = replay.sample_indices(cfg.batch_size)
                loss, prio = hl.step(ring, torch.from_numpy(idx).to(dev), None, cfg.batch_size)
                steps += 1
                if steps % args.report_every == 0:
                    c = p.selfplay_counters()
                    de, ds = c['episodes'] - last['episodes'], c['episode_steps'] - last['episode_steps']
                    last = dict(episodes=c['episodes'], episode_steps=c['episode_steps'])
                    rec = dict(train_steps=steps, env_steps=c['env_steps'], episodes_finished=de, mean_episode_length=(ds / de) if de else None,
                               loss=float(loss), replay=replay.size, seconds=round(time.time() - t0, 1))
                    log.append(rec)
                    print(json.dumps(rec), flush=True)
                continue
            batch, idx, w = replay.sample_tensors(cfg.batch_size)
            if args.eager_learner:
                loss, prio = learner.train_step(cfg, net, opt, sched, dev, batch, w)
                replay.update_priorities(idx, prio)
            else:  # the whole update as one HIP graph; uniform replay (the launchers' default) ignores priorities: no host read-back
                if graphed is None:
                    graphed = learner.GraphedTrainStep(cfg, net, opt, dev, cfg.batch_size, (4, 5), cfg.unroll_steps, 2)
                loss, prio = graphed(batch, w)
                sched.step()
            steps += 1
            if steps % args.report_every == 0:
                c = p.selfplay_counters()
                de, ds = c['episodes'] - last['episodes'], c['episode_steps'] - last['episode_steps']
                last = dict(episodes=c['episodes'], episode_steps=c['episode_steps'])
                rec = dict(train_steps=steps, env_steps=c['env_steps'], episodes_finished=de, mean_episode_length=(ds / de) if de else None,
                           loss=float(loss), replay=replay.size, seconds=round(time.time() - t0, 1))
                log.append(rec)
                print(json.dumps(rec), flush=True)
        net.eval()
        p.load_state_dict(net.state_dict())  # actor <- learner (the reference does this every checkpoint_interval steps)
        if args.eval_every and steps < args.train_steps and steps // args.eval_every > (steps - args.updates_per_iter) // args.eval_every:
            evals.append(dict(train_steps=steps, eval_episode_lengths=evaluate()))
            print(json.dumps(evals[-1]), flush=True)
    lengths = evaluate()
    evals.append(dict(train_steps=steps, eval_episode_lengths=lengths))
    print(json.dumps(dict(eval_episode_lengths=lengths)), flush=True)
    if args.out:
        json.dump(dict(args=vars(args), log=log, eval_episode_lengths=lengths, evals=evals), open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
