#!/usr/bin/env python3
"""Diagnostic: GPU search vs oracle on a network TRAINED on the box (trained value ranges are where near-ties live; checkpoints
cannot travel).  Trains CartPole for --train-steps, then searches states of evaluation episodes with every kernel variant and the
oracle, deterministic and sampled, and plays deterministic evaluation episodes with each variant.
    python tests/trained_parity.py --train-steps 4000"""
import argparse
import json
import os
import sys
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, os.path.join(REPO, 'oracle'))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def train(steps, envs=128, seed=1):
    from muzero_amd import learner
    from muzero_amd import planner as pl
    from muzero_amd.config import make_classic_config
    from muzero_amd.network import MuZeroMLPNet
    from muzero_amd.replay import PrioritizedReplay

    torch.manual_seed(seed)
    dev = torch.device('cuda', 0)
    cfg = make_classic_config(num_training_steps=steps, batch_size=128, min_replay_size=5000, use_tensorboard=False)
    net = MuZeroMLPNet((4, 5), 2, cfg.num_planes, cfg.value_support_size, cfg.reward_support_size, cfg.hidden_dim).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=cfg.lr_init, weight_decay=cfg.weight_decay)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=cfg.lr_milestones, gamma=cfg.lr_decay_rate)
    replay = PrioritizedReplay(50000, 0.0, 0.0, np.random.RandomState(seed), device='cuda')
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), cfg, num_envs=envs, seed=seed), 0)
    net.eval()
    p.load_state_dict(net.state_dict())
    p.attach_replay(replay, cfg, obs_shape=(4, 5))
    p.selfplay_reset(pl.ENV_CARTPOLE)
    n = 0
    while n < steps:
        p.selfplay_step(1.0, 8)
        if replay.size < cfg.min_replay_size:
            continue
        net.train()
        for _ in range(16):
            batch, idx, w = replay.sample_tensors(cfg.batch_size)
            learner.train_step(cfg, net, opt, sched, dev, batch, w)
            n += 1
        net.eval()
        p.load_state_dict(net.state_dict())
    c = p.selfplay_counters()
    p.detach_replay()
    p.close()
    return cfg, net.cpu()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--train-steps', type=int, default=4000)
    ap.add_argument('--ckpt', default='/tmp/mz_trained.pt')
    ap.add_argument('--child', default='')
    args = ap.parse_args()
    from muzero_amd.config import make_classic_config
    from muzero_amd.network import MuZeroMLPNet
    from muzero_amd.games import CartPoleEnv
    from muzero_amd import planner as pl

    if not args.child:
        cfg, net = train(args.train_steps)
        torch.save(net.state_dict(), args.ckpt)
        for name, env in (('default', {}), ('MZ_HWX=0', {'MZ_HWX': '0'}), ('generic', {'MZ_FORCE_GENERIC': '1'}), ('nohw', {'MZ_LIB': os.path.join(REPO, 'muzero_amd/lib/libmz_nohw.so')})):
            if 'MZ_LIB' in env and not os.path.exists(env['MZ_LIB']):
                continue
            e = dict(os.environ, **env)
            subprocess.run([sys.executable, os.path.abspath(__file__), '--child', name, '--ckpt', args.ckpt], env=e, check=False)
        return
    if os.environ.get('MZ_LIB'):
        pl.LIB_PATH = os.environ['MZ_LIB']
    import oracle
    from test_oracle_nets import _oracle_net
    from muzero_amd import mcts

    cfg = make_classic_config(use_tensorboard=False)
    net = MuZeroMLPNet((4, 5), 2, cfg.num_planes, cfg.value_support_size, cfg.reward_support_size, cfg.hidden_dim)
    net.load_state_dict(torch.load(args.ckpt))
    net.eval()
    dev = torch.device('cuda', 0)
    # 1. deterministic evaluation episodes through mcts.uct_search (what examples/train_cartpole.py does)
    lengths = []
    states = []
    for ep in range(3):
        env = CartPoleEnv(4, seed=1000 + ep)
        obs, done = env.reset(), False
        while not done:
            if len(states) < 600:
                states.append(obs.copy())
            action, *_ = mcts.uct_search(obs, net, dev, cfg, 0.0, env.actions_mask, 1, 1, deterministic=True)
            obs, _, done, _ = env.step(action)
        lengths.append(env.steps)
    # 2. the same states, batched, against the oracle: deterministic and sampled with injected draws
    B, S = len(states), cfg.num_simulations
    obs = np.stack(states).astype(np.float32)
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), cfg, num_envs=B, seed=5), 0)
    p.load_state_dict(net.state_dict())
    onet = _oracle_net(oracle, net, 'mlp')
    ocfg = oracle.make_config(2, S, cfg.discount, False, None, cfg.root_dirichlet_alpha, cfg.root_exploration_eps)
    rs = np.random.RandomState(3)
    mask = np.ones((B, 2), bool)
    out = {}
    for det in (True, False):
        rng = dict(noise=None if det else rs.dirichlet(np.full(2, cfg.root_dirichlet_alpha), size=B), u_tie=rs.rand(B, 4 * S + 8), u_final=rs.rand(B))
        r = p.search(obs, mask, 1, 1, 0.0 if det else 1.0, det, **rng)
        o = oracle.uct_search_batch(ocfg, onet, obs, mask.astype(np.uint8), 1, 1, 0.0 if det else 1.0, det, **rng)
        out['det' if det else 'sampled'] = dict(visits_equal=int((r['visits'] == o['visits']).all(axis=1).sum()), action_equal=int((r['action'] == o['action']).sum()),
                                                  root_equal=int((r['root_value'] == o['root_value']).sum()), n=B,
                                                  root_range=[float(o['root_value'].min()), float(o['root_value'].max())])
    print(json.dumps({'variant': args.child, 'eval_lengths': lengths, 'vs_oracle': out}), flush=True)


if __name__ == '__main__':
    main()
