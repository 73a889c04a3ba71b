#!/usr/bin/env python3
"""bench.py -- self-play planning throughput on MI355X (BASELINE.json metric: self-play env-steps/sec & MCTS sims/sec).

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1], "C2"): CartPole-v1, 50 simulations per move, 4096 parallel environments per GPU,
MuZeroMLPNet 512/64/31 (seeded random-init weights, synthetic data), device-resident CartPole environments with
auto-reset, on-device Philox randomness, temperature 1.0.  One "step" = one lock-step self-play move for all
environments of a rank = ONE kernel launch: temperature + record, root inference, 50 x {select, dynamics/reward/value inference,
expand, backup}, play policy + action sample, env.step + record + auto-reset.  All inputs are resident in HBM.

Multi-GPU: environments are independent (one planner per GPU, envs sharded by rank, no data-path collective);
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the elapsed time.  scaling = "weak".

Output: ONE JSON line on rank 0 with the contract fields plus `roofline` (dominant kernel k_search, fp32 MFMA bound,
measured live with HIP events on the planner's stream) and `cpu_baseline` (the CPU oracle -- a C port of the reference
algorithm -- timed on the host cores on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

# CartPole MLP (C2): algorithmic MACs per simulation of the reference-faithful search (the recurrent policy head is
# dead compute in the reference, mcts.py:386 uses the root prior, so it is not evaluated): SURVEY 8d.
MAC_TRANSITION = 66 * 512 + 512 * 64
MAC_REWARD = 64 * 512 + 512 * 31
MAC_VALUE = 64 * 512 + 512 * 31
MAC_POLICY = 64 * 512 + 512 * 2
MAC_REPRESENT = 20 * 512 + 512 * 64
FLOP_PER_SIM = 2 * (MAC_TRANSITION + MAC_REWARD + MAC_VALUE)          # 327 680
# root: representation + policy head only -- the search discards the root's value (mcts.py:356-367) and the kernel does not
# evaluate that head (mz_search_fast.h: mlp_initial_tile(..., want_value = false)), so it is not counted
FLOP_PER_ROOT = 2 * (MAC_REPRESENT + MAC_POLICY)                      # 153 600
PEAK_FP32_MFMA_TFLOPS = 157.3                                         # MI355X_MICROARCH.md, v_mfma_f32_*_f32 dense peak


def cpu_baseline(num_sims, sample_envs, budget_s=20.0):
    """The oracle (oracle/mz_oracle.c, a scalar C port of mcts.py + network.py, one env per host thread with batch-1
    inference like the reference's one-actor-per-process layout) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import oracle  # test infrastructure, used here only as the reported CPU baseline
    from helpers import build_mlp, mlp_case

    net = build_mlp(mlp_case('cartpole'))
    sd = {k: v.numpy() for k, v in net.state_dict().items()}
    onet = oracle.Net.mlp(sd, 20, 2, 512, 64, 31, 31)
    cfg = oracle.make_config(2, num_sims, 0.997, False, None, 0.25, 0.25)
    cores = os.cpu_count() or 1
    rs = np.random.RandomState(3)

    def run(n_envs, threads):
        obs = rs.uniform(-0.05, 0.05, size=(n_envs, 4, 5)).astype(np.float32)
        obs[:, :, 4] = 0.5
        noise = rs.dirichlet(np.full(2, 0.25), size=n_envs)
        t0 = time.perf_counter()
        oracle.uct_search_batch(cfg, onet, obs, np.ones((n_envs, 2), np.uint8), 1, 1, 1.0, False, noise=noise,
                                u_tie=rs.rand(n_envs, 4 * num_sims + 8), u_final=rs.rand(n_envs), num_threads=threads)
        return n_envs * num_sims / (time.perf_counter() - t0)

    # pick the thread count that runs fastest on this host (SMT siblings can hurt), on a small probe
    best_threads, best_rate = 1, run(8, 1)
    for t in sorted({max(1, cores // 2), cores}):
        r = run(8 * t, t)
        if r > best_rate:
            best_threads, best_rate = t, r
    n_envs = int(max(best_threads * 8, min(sample_envs, best_rate * budget_s / num_sims)))
    rate = run(n_envs, best_threads)
    return dict(value=rate, unit='sims/s', cores=best_threads, kind='port',
                sample=f'{n_envs} CartPole roots x {num_sims} simulations, oracle/mz_oracle.c, {best_threads} OpenMP threads '
                       f'of {cores} host CPUs, batch-1 inference per env')


# Conv workloads (BASELINE.json configs[3] / configs[4]; not the headline bench line): name -> (case tuple for
# tests/helpers.build_conv, envs per GPU, sims per move, env kind, search kwargs)
CONV_WORKLOADS = {
    'c4': (('c4', 'atari', (8, 96, 96), 6, 8, 128, 61, 61, 41), 512, 50, 'synthetic', dict(discount=0.997, root_dirichlet_alpha=0.25)),
    'c5': (('c5', 'board', (9, 15, 15), 226, 8, 128, 1, 1, 42), 256, 200, 'gomoku',
           dict(discount=1.0, is_board_game=True, known_bounds=(-1.0, 1.0), root_dirichlet_alpha=0.03)),
}


def conv_flops(case):
    """Algorithmic FLOPs (2 x MAC) of one simulation (dynamics + prediction towers, reward + value heads; the recurrent
    policy head is dead compute in the reference search) and of one root inference, SURVEY 8d."""
    _, kind, (c, h, w), A, R, P, Sv, Sr, _ = case
    hh, hw = (6, 6) if kind == 'atari' else (h, w)
    px = hh * hw
    res = R * 2 * P * P * 9 * px
    sim = (P + A) * P * 9 * px + res + (P * px + px * Sr) + res + (P * px + px * Sv)
    if kind == 'atari':
        rep = c * 128 * 9 * 48 * 48 + 4 * 128 * 128 * 9 * 48 * 48 + 128 * P * 9 * 24 * 24 + 4 * P * P * 9 * 24 * 24 + 4 * P * P * 9 * 12 * 12
    else:
        rep = c * P * 9 * px + res
    root = rep + res + (2 * P * px + 2 * px * A)  # (the root's value head is discarded by the search, mcts.py:356-367: not counted)
    return 2 * sim, 2 * root


def conv_cpu_baseline(name, case, net, kw, full_sims):
    """The oracle (scalar C port, one env per OpenMP thread, batch-1 inference like the reference's actors) on a bounded
    sample of the same network.  MLP nets: whole searches.  Conv nets: one root per thread searched with 2 and with 6
    simulations; the difference gives the cost of a simulation, the rest the root inference, and the reported rate is that of
    whole moves of `full_sims` simulations (root inference included once per move, as in the reference)."""
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import oracle  # test infrastructure, used here only as the reported CPU baseline

    onet = oracle.Net.from_module(net, 'mlp' if name == 'c3' else 'conv')
    A = case[3]
    threads = max(1, (os.cpu_count() or 2) // 2)
    rs = np.random.RandomState(5)
    board = bool(kw.get('is_board_game', False))

    def run(n_envs, sims):
        obs = rs.uniform(0, 1, size=(n_envs,) + tuple(case[2])).astype(np.float32)
        cfg = oracle.make_config(A, sims, kw['discount'], board, kw.get('known_bounds'), kw['root_dirichlet_alpha'], 0.25)
        noise = rs.dirichlet(np.full(A, kw['root_dirichlet_alpha']), size=n_envs)
        t0 = time.perf_counter()
        oracle.uct_search_batch(cfg, onet, obs, np.ones((n_envs, A), np.uint8), 1, 2 if board else 1, 1.0, False, noise=noise,
                                u_tie=rs.rand(n_envs, 4 * sims + 8), u_final=rs.rand(n_envs), num_threads=threads)
        return time.perf_counter() - t0

    if name == 'c3':
        n = threads * 8
        dt = run(n, full_sims)
        return dict(value=n * full_sims / dt, unit='sims/s', cores=threads, kind='port',
                    sample=f'{n} roots x {full_sims} simulations, oracle/mz_oracle.c, {threads} OpenMP threads, {dt:.1f} s')
    t2, t6 = run(threads, 2), run(threads, 6)
    t_sim = max(1e-9, (t6 - t2) / 4.0)
    t_root = max(0.0, t2 - 2.0 * t_sim)
    rate = threads * full_sims / (t_root + full_sims * t_sim)
    return dict(value=rate, unit='sims/s', cores=threads, kind='port',
                sample=f'{threads} roots (one per OpenMP thread) searched with 2 and 6 simulations, oracle/mz_oracle.c: {t_sim:.3f} s per simulation, '
                       f'{t_root:.2f} s per root inference per thread; rate of whole {full_sims}-simulation moves')


DOMINANT = {'c3': 'k_search_fast<256', 'c4': 'k_res_tower<', 'c5': 'k_conv3x3<15, 1, true, 15, false>'}  # dominant kernel of each workload (substring of its rocprof name)


def measure_workload(args, name, rank, local_rank, world, torch, dist, red_dev, steps, warmup, preheat, with_cpu, with_sustained):
    """One of the other BASELINE.json configs (c3 / c4 / c5) measured exactly like the headline: `warmup` untimed lock-step
    moves, then `steps` timed ones between barriers, MAX over ranks; HIP-event pairs on the planner stream around each move's
    kernel sequence give the roofline figure.  Returns the record (rank 0; None elsewhere)."""
    from helpers import build_conv, build_mlp, mlp_case
    from muzero_amd import planner as pl

    if name == 'c3':  # BASELINE.json configs[2]: TicTacToe MLP 256/64, MSE heads, two-player backup, 25 sims, 4096 envs
        envs, sims, env_kind = 4096, 25, 'tictactoe'
        kw = dict(discount=1.0, is_board_game=True, known_bounds=(-1.0, 1.0), root_dirichlet_alpha=0.25)
        net = build_mlp(mlp_case('tictactoe'))
        case = ('c3', 'mlp', (9, 3, 3), 10, 0, 256, 1, 1, 13)
    else:
        case, envs, sims, env_kind, kw = CONV_WORKLOADS[name]
        net = build_conv(case)
    B = args.envs or envs
    S = args.sims or sims
    cfg = pl.make_mz_config(net.planner_spec(), None, num_envs=B, seed=1000 + rank, num_simulations=S, root_exploration_eps=0.25, **kw)
    p = pl.Planner(cfg, local_rank)
    p.load_state_dict(net.state_dict())
    p.selfplay_reset({'gomoku': pl.ENV_GOMOKU, 'tictactoe': pl.ENV_TICTACTOE}.get(env_kind, pl.ENV_SYNTHETIC))
    T = -1.0 if env_kind in ('gomoku', 'tictactoe') else 1.0

    def sync():
        p.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if preheat and name == 'c3':  # see the C2 path: out of the idle clocks (one move of the conv workloads is long enough by itself)
        p.selfplay_step(T, preheat)
    if warmup:
        p.selfplay_step(T, warmup)
    sync()
    p.profile_begin()
    t0 = time.perf_counter()
    p.selfplay_step(T, steps)
    sync()
    elapsed = time.perf_counter() - t0
    prof = p.profile_end()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # C3: transition 35 328 + reward 16 640 + value 16 640 MAC per simulation; root 37 120 + 18 944 (representation +
    # policy; the root's value head is discarded by the search and not evaluated by the kernel: not counted)
    f_sim, f_root = (2 * 68608, 2 * 56064) if name == 'c3' else conv_flops(case)
    flop_per_move = B * (S * f_sim + f_root)
    sustained = sustained_leg(p, T, 1e3 * elapsed / steps, flop_per_move) if with_sustained else None
    rec = None
    if rank == 0:
        sims_per_s = world * B * S * steps / elapsed
        ms = prof['search_kernel_ms'] / max(1, prof['search_kernel_launches'])
        traffic, traffic_src = profiled_traffic(name, DOMINANT[name]) if (B == envs and S == sims) else (None, 'non-default workload size')
        achieved = flop_per_move / (ms * 1e-3) / 1e12
        ms_step = 1e3 * elapsed / steps
        cpu = conv_cpu_baseline(name, case, net, kw, S) if (world == 1 and with_cpu) else None
        rec = {
            'cpu_baseline': cpu,
            'metric': 'self-play MCTS sims/sec (env-steps/sec = value / sims_per_move)', 'value': sims_per_s, 'unit': 'sims/s', 'n_gpus': world,
            'steps': steps, 'warmup': warmup, 'ms_per_step': ms_step, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{name.upper()}: {case[1]} net {case[5]} planes / {case[4]} blocks, obs {case[2]}, A={case[3]}, {S} sims/move, '
                                   f'{B} envs per MI355X, {env_kind} device env, ' + ('LDS-resident trees' if name == 'c3' else 'HBM-resident trees'),
                       'envs_per_gpu': B, 'sims_per_move': S, 'parallelism': f'env-sharded x{world}', 'weights': 'seeded random init'},
            'env_steps_per_sec': sims_per_s / S,
            'roofline': {'bound': 'mfma', 'kernel': 'mz::' + DOMINANT[name] + ('...> (the one kernel of a move: search + env step fused)' if name == 'c3'
                                                                               else '...> (dominant kernel; the whole per-move kernel sequence is timed)'),
                         'dispatch': p.describe(), 'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP32_MFMA_TFLOPS,
                         'frac_step': flop_per_move / (ms_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 'traffic': traffic,
                         'traffic_unit': 'bytes of HBM per launch of the dominant kernel', 'traffic_source': traffic_src,
                         'avg_move_ms': ms, 'flop_per_move': flop_per_move, 'flop_per_sim': f_sim,
                         'timed_with': 'hipEvent pairs on the planner stream around each move\'s kernel sequence'},
            'sustained': sustained,
        }
    p.close()
    del p, net
    torch.cuda.empty_cache()
    return rec


def run_conv_workload(args, name, rank, local_rank, world, torch, dist, red_dev='cuda', backend='nccl'):
    rec = measure_workload(args, name, rank, local_rank, world, torch, dist, red_dev, args.steps, args.warmup, args.preheat,
                           not args.no_cpu_baseline, not args.no_sustained)
    if rank == 0:
        rec['distributed'] = dist_record(dist, world, backend)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def dist_record(dist, world, backend):
    """What the process group really was: the first multi-GPU run is also the first time RCCL sees N ranks."""
    if world > 1:
        return {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'requested_backend': backend,
                'note': 'backend "nccl" is RCCL on ROCm; used only for the barriers and the MAX of the elapsed time (no data-path collective)'}
    return {'backend': None, 'world_size': 1}


# the other BASELINE.json configs inside the default (headline) run: (timed moves, warm-up moves)
CONFIG_LEGS = {'c3': (200, 20), 'c4': (4, 1), 'c5': (2, 1)}


def configs_leg(args, rank, local_rank, world, torch, dist, red_dev):
    """BASELINE.json lists five configs; the headline line is C2.  This leg times a few lock-step moves of C3, C4 and C5 in the
    same driver-run process (same measurement as `--workload cN`, fewer moves; full-size batches, all simulations), so every
    config has a driver-timed figure with its own roofline sub-record every round."""
    import types

    out = {}
    only = os.environ.get('MZ_BENCH_CONFIGS')  # diagnostics: a comma list of the legs to run
    for name, (steps, warm) in CONFIG_LEGS.items():
        if only and name not in only.split(','):
            continue
        a = types.SimpleNamespace(envs=0, sims=0)
        r = measure_workload(a, name, rank, local_rank, world, torch, dist, red_dev, steps, warm, 0, False, False)
        if rank == 0:
            rf = r['roofline']
            out[name] = {'workload': r['config']['workload'], 'value': r['value'], 'unit': r['unit'], 'env_steps_per_sec': r['env_steps_per_sec'],
                         'steps': steps, 'warmup': warm, 'ms_per_step': r['ms_per_step'], 'n_gpus': world,
                         'roofline': {k: rf[k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'frac_step', 'traffic', 'traffic_source',
                                                         'avg_move_ms', 'flop_per_move')}}
    return out


PROFILE_ROUND = 'round6'


def lunar_leg(rank, local_rank, world, torch, dist, red_dev, steps=100, warm=20, preheat=100):
    """The north-star's "4 actions" wording: a LunarLander-v2-SHAPED search workload (MuZeroMLPNet 512 / 64 / 31 on (4, 9) observations,
    A = 4, config.py:170-201; the reference's shipped LunarLander checkpoint has these shapes), 4096 envs x 50 simulations.  Box2D is an
    absent dependency, so the env is the synthetic stand-in (fresh U[0,1) observations, reward 0, 1000-step episodes: MZ_ENV_SYNTHETIC);
    the search runs the general-action-count build of the tuned kernel, k_search_fast<512, 2, 2, *, 0, ...>."""
    from helpers import build_mlp, mlp_case
    from muzero_amd import planner as pl

    net = build_mlp(mlp_case('lunar'))
    B, S = 4096, 50
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=B, seed=2000 + rank, num_simulations=S, discount=0.997,
                                     root_dirichlet_alpha=0.25, root_exploration_eps=0.25), local_rank)
    p.load_state_dict(net.state_dict())
    p.selfplay_reset(pl.ENV_SYNTHETIC)

    def sync():
        p.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    p.selfplay_step(1.0, preheat + warm)
    sync()
    p.profile_begin()
    t0 = time.perf_counter()
    p.selfplay_step(1.0, steps)
    sync()
    elapsed = time.perf_counter() - t0
    prof = p.profile_end()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    lunar_dispatch = p.describe()
    p.close()
    # transition (64 + 4) x 512 + 512 x 64, reward and value 64 x 512 + 512 x 31 each; root: representation 36 x 512 + 512 x 64 + policy 64 x 512 + 512 x 4
    f_sim = 2 * ((68 * 512 + 512 * 64) + 2 * (64 * 512 + 512 * 31))
    f_root = 2 * ((36 * 512 + 512 * 64) + (64 * 512 + 512 * 4))
    flop = B * (S * f_sim + f_root)
    k_ms = prof['search_kernel_ms'] / max(1, prof['search_kernel_launches'])
    ms_step = 1e3 * elapsed / steps
    traffic, traffic_src = profiled_traffic('lunar', 'k_search_fast<512')
    return {'workload': 'LunarLander-shaped: MLP 512/64/31, obs (4, 9), A=4, 50 sims/move, 4096 envs per MI355X, synthetic observations (Box2D absent)',
            'value': world * B * S * steps / elapsed, 'unit': 'sims/s', 'env_steps_per_sec': world * B * steps / elapsed, 'steps': steps, 'warmup': warm,
            'ms_per_step': ms_step, 'n_gpus': world,
            'roofline': {'bound': 'mfma', 'kernel': lunar_dispatch,
                         'achieved': flop / (k_ms * 1e-3) / 1e12, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': flop / (k_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         'frac_step': flop / (ms_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 'traffic': traffic, 'traffic_source': traffic_src,
                         'avg_move_ms': k_ms, 'flop_per_move': flop}}


def learner_leg(rank, local_rank, world, torch, dist, backend, batches=(128, 4096), iters=100):
    """Row f2, the consumer of the self-play stream: one update of the reference's learner (calc_loss + backward + Adam, pipeline.py:238-255)
    on the hand-written kernels (hip_learner.HipLearner), classic-control net, unroll 5, batch gathered from an HBM ring by index.  With
    N > 1 ranks the flat gradient (0.97 MB) is averaged by one RCCL all-reduce per update.  Algorithmic FLOPs: 6 x MAC (forward + twice that
    for the backward pass)."""
    from muzero_amd.config import make_classic_config
    from muzero_amd.hip_learner import HipLearner
    from muzero_amd.network import MuZeroMLPNet

    dev = torch.device('cuda', local_rank)
    cfg = make_classic_config(use_tensorboard=False)
    K, A, cap = cfg.unroll_steps, 2, 65536
    g = torch.Generator(device='cpu').manual_seed(7 + rank)
    ring = dict(state=(torch.rand(cap, 20, generator=g) * 2 - 1).to(dev), action=torch.randint(0, A, (cap, K), generator=g).to(torch.int8).to(dev),
                pi_prob=torch.full((cap, K, A), 0.5, device=dev), value=(torch.rand(cap, K, generator=g) * 50).to(dev), reward=torch.ones(cap, K, device=dev))
    mac = 20 * 512 + 512 * 64 + K * ((66 * 512 + 512 * 64) + 2 * (64 * 512 + 512 * 31) + (64 * 512 + 512 * A))
    rows = []
    for B in batches:
        torch.manual_seed(0)
        net = MuZeroMLPNet((4, 5), A, cfg.num_planes, cfg.value_support_size, cfg.reward_support_size, cfg.hidden_dim).to(dev)
        hl = HipLearner(net, dev, K, B, lr=cfg.lr_init, weight_decay=cfg.weight_decay, milestones=cfg.lr_milestones, gamma=cfg.lr_decay_rate)
        idx = torch.randint(0, cap, (B,), generator=g).to(dev)
        ar = world > 1 and backend == 'nccl'
        for _ in range(20):
            hl.step(ring, idx, None, B, allreduce=ar)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            hl.step(ring, idx, None, B, allreduce=ar)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / iters
        tf = 6.0 * mac * B / (ms * 1e-3) / 1e12
        rows.append({'batch_per_gpu': B, 'ms_per_update': ms, 'samples_per_sec': world * B / (ms * 1e-3), 'achieved_tflops_per_gpu': tf,
                     'mfma_frac': tf / PEAK_FP32_MFMA_TFLOPS, 'gradient_allreduce': 'RCCL, one flat %.2f MB bucket per update' % (hl.total * 4 / 1e6) if ar else None})
        if world > 1 and ar:  # after the all-reduced updates every rank must hold bit-identical weights
            mine = hl.params.clone()
            ref = mine.clone()
            dist.broadcast(ref, src=0)
            same = torch.tensor([1 if torch.equal(mine, ref) else 0], device=dev)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            rows[-1]['ranks_hold_identical_weights'] = bool(int(same))
            assert bool(int(same)), 'data-parallel learner: ranks diverged after all-reduced updates'
        hl.close()
    return {'what': 'hip_learner.HipLearner: loss + backward + Adam + operand re-pack as gfx950 kernels, MuZeroMLPNet 512/64/31, unroll 5', 'rows': rows,
            'flop_per_sample': 6.0 * mac, 'conv': conv_learner_leg(rank, local_rank, world, torch, dist, backend),
            'atari': atari_learner_leg(rank, local_rank, world, torch, dist, backend)}


def conv_learner_leg(rank, local_rank, world, torch, dist, backend, batch=128, iters=10):
    """Row f2, the conv half (round 5): one update of the C5 network -- MuZeroBoardGameNet 15 x 15, 128 planes, 8 residual blocks, 226 actions,
    unroll 5, the reference's Gomoku batch size 128 (config.py:113-119) -- on the kernels of csrc/mz_learn_conv.h: train-mode BatchNorm towers,
    heads, losses, backward, Adam, operand re-pack.  Algorithmic FLOPs: hip_learner.conv_learner_flops (forward + weight gradient of every 3 x 3
    conv + data gradient wherever an input needs one).  With N > 1 ranks the 30.4 MB flat gradient is averaged by one RCCL all-reduce per update."""
    from muzero_amd.config import make_gomoku_config
    from muzero_amd.hip_learner import HipLearner, conv_learner_flops
    from muzero_amd.network import MuZeroBoardGameNet

    dev = torch.device('cuda', local_rank)
    cfg = make_gomoku_config(use_tensorboard=False)
    N, K, cap = 15, cfg.unroll_steps, 1024
    A, shape = N * N + 1, (9, N, N)
    g = torch.Generator(device='cpu').manual_seed(17 + rank)
    ring = dict(state=torch.randint(0, 2, (cap, 9 * N * N), generator=g).to(torch.int8).to(dev), action=torch.randint(0, A, (cap, K), generator=g).to(torch.int16).to(dev),
                pi_prob=torch.full((cap, K, A), 1.0 / A, device=dev), value=(torch.rand(cap, K, generator=g) * 2 - 1).to(dev),
                reward=(torch.rand(cap, K, generator=g) * 2 - 1).to(dev))
    torch.manual_seed(0)
    net = MuZeroBoardGameNet(shape, A, cfg.num_res_blocks, cfg.num_planes).to(dev)
    hl = HipLearner(net, dev, K, batch, lr=cfg.lr_init, weight_decay=cfg.weight_decay, milestones=cfg.lr_milestones, gamma=cfg.lr_decay_rate)
    idx = torch.randint(0, cap, (batch,), generator=g).to(dev)
    ar = world > 1 and backend == 'nccl'
    for _ in range(3):
        hl.step(ring, idx, None, batch, allreduce=ar)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        hl.step(ring, idx, None, batch, allreduce=ar)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters
    flop = conv_learner_flops(shape, A, cfg.num_res_blocks, cfg.num_planes, K)
    tf = flop * batch / (ms * 1e-3) / 1e12
    rec = {'what': 'hip_learner.HipLearner on MuZeroBoardGameNet 15x15 / 128 planes / 8 blocks / A=226, unroll 5 (the C5 net): loss + backward + Adam as gfx950 kernels',
           'batch_per_gpu': batch, 'ms_per_update': ms, 'samples_per_sec': world * batch / (ms * 1e-3), 'flop_per_sample': flop, 'achieved_tflops_per_gpu': tf,
           'mfma_frac': tf / PEAK_FP32_MFMA_TFLOPS, 'parameters': int(hl.total),
           'gradient_allreduce': 'RCCL, one flat %.1f MB bucket per update' % (hl.total * 4 / 1e6) if ar else None,
           'pytorch_rocm_same_update_ms': {'eager_miopen': 73.6, 'hip_graph_miopen': 71.9, 'source': 'tools/conv_learner_bench.py, round 4 (DESIGN 4b)'}}
    if world > 1 and ar:
        mine = hl.params.clone()
        ref = mine.clone()
        dist.broadcast(ref, src=0)
        same = torch.tensor([1 if torch.equal(mine, ref) else 0], device=dev)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        rec['ranks_hold_identical_weights'] = bool(int(same))
    hl.close()
    return rec


def atari_learner_leg(rank, local_rank, world, torch, dist, backend, batch=128, iters=6):
    """Row f2, the Atari net (round 5): one update of make_atari_config's network (config.py:132-141: 128 planes, 8 blocks, supports 61, batch 128,
    unroll 5) on 4 x 96 x 96 frame stacks with 6 actions -- the tile path of csrc/mz_learn_conv.h.  Algorithmic FLOPs:
    hip_learner.atari_learner_flops (the halo positions the tiles convolve are not counted)."""
    from muzero_amd.config import make_atari_config
    from muzero_amd.hip_learner import HipLearner, atari_learner_flops
    from muzero_amd.network import MuZeroAtariNet

    dev = torch.device('cuda', local_rank)
    cfg = make_atari_config(use_tensorboard=False)
    K, cap, A, shape = cfg.unroll_steps, 256, 6, (4, 96, 96)
    g = torch.Generator(device='cpu').manual_seed(23 + rank)
    ring = dict(state=torch.rand(cap, 4 * 96 * 96, generator=g).to(dev), action=torch.randint(0, A, (cap, K), generator=g).to(torch.int8).to(dev),
                pi_prob=torch.full((cap, K, A), 1.0 / A, device=dev), value=(torch.rand(cap, K, generator=g) * 20 - 10).to(dev),
                reward=(torch.rand(cap, K, generator=g) * 2 - 1).to(dev))
    torch.manual_seed(0)
    net = MuZeroAtariNet(shape, A, cfg.num_res_blocks, cfg.num_planes, cfg.value_support_size, cfg.reward_support_size).to(dev)
    hl = HipLearner(net, dev, K, batch, lr=cfg.lr_init, weight_decay=cfg.weight_decay, milestones=cfg.lr_milestones, gamma=cfg.lr_decay_rate)
    idx = torch.randint(0, cap, (batch,), generator=g).to(dev)
    ar = world > 1 and backend == 'nccl'
    for _ in range(2):
        hl.step(ring, idx, None, batch, allreduce=ar)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        hl.step(ring, idx, None, batch, allreduce=ar)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters
    flop = atari_learner_flops(shape, A, cfg.num_res_blocks, cfg.num_planes, K)
    tf = flop * batch / (ms * 1e-3) / 1e12
    rec = {'what': 'hip_learner.HipLearner on MuZeroAtariNet 4x96x96 / 128 planes / 8 blocks / A=6 / supports 61, unroll 5 (make_atari_config): loss + backward + Adam as gfx950 kernels',
           'batch_per_gpu': batch, 'ms_per_update': ms, 'samples_per_sec': world * batch / (ms * 1e-3), 'flop_per_sample': flop, 'achieved_tflops_per_gpu': tf,
           'mfma_frac': tf / PEAK_FP32_MFMA_TFLOPS, 'parameters': int(hl.total),
           'pytorch_rocm_same_update_ms': {'eager_miopen': 45.8, 'hip_graph_miopen': 43.3, 'source': 'tools/conv_learner_bench.py --atari, round 5 (DESIGN 4c)'}}
    hl.close()
    return rec


def train_leg(rank, local_rank, world, torch, dist, seconds=2.0, batch=4096, moves=8):
    """The whole loop on one GPU at the headline workload's size -- run_self_play + run_training of the reference (pipeline.py:41-286) as the
    planner (C2: 4096 CartPole envs, 50 sims, device epilogue into a 1 M-item HBM replay) and the HIP learner taking turns on the device, in
    event order: `moves` lock-step moves, then one update per move of `batch` samples drawn from the replay (about one sampled item per env
    step; the reference's classic run trains ~11 per env step at 340 env-steps/s), the planner re-fed with the learner's weights every 16
    iterations.  Reports what the pair sustains together."""
    import types

    from muzero_amd import learner
    from muzero_amd import planner as pl
    from muzero_amd.config import make_classic_config
    from muzero_amd.network import MuZeroMLPNet
    from muzero_amd.replay import PrioritizedReplay

    dev = torch.device('cuda', local_rank)
    cfg = make_classic_config(batch_size=batch, min_replay_size=batch, use_tensorboard=False)
    torch.manual_seed(3 + rank)
    net = MuZeroMLPNet((4, 5), 2, cfg.num_planes, cfg.value_support_size, cfg.reward_support_size, cfg.hidden_dim).to(dev)
    hl = learner.make_hip_learner(cfg, net, dev)
    B = 4096
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), cfg, num_envs=B, seed=3000 + rank), local_rank)
    net.eval()
    p.load_state_dict(net.state_dict())
    rp = PrioritizedReplay(1 << 20, 0.0, 0.0, np.random.RandomState(11 + rank), device='cuda')
    # make_classic_config's own flush rule (acc_seq_length 9999: items at episode ends only, config.py:198)
    p.attach_replay(rp, types.SimpleNamespace(is_board_game=False, acc_seq_length=cfg.acc_seq_length, unroll_steps=cfg.unroll_steps, td_steps=cfg.td_steps,
                                              discount=cfg.discount), obs_shape=(4, 5))
    p.selfplay_reset(pl.ENV_CARTPOLE)
    p.selfplay_step(1.0, 240)  # past the first episode ends: the replay is filling
    p.synchronize()

    sampler = rp.device_sampler(seed=23 + rank)  # draws on the device from the committed counter the epilogue publishes: no host read per update

    def iteration():
        p.selfplay_step(1.0, moves)
        p.synchronize()
        for _ in range(moves):
            idx, _, ring = sampler.sample(batch)
            hl.step(ring, idx, None, batch, allreduce=False)

    for _ in range(3):
        iteration()
    torch.cuda.synchronize()
    it, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        iteration()
        it += 1
        if it % 16 == 0:
            p.load_state_dict(net.state_dict())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss = float(hl.loss)
    p.close()
    hl.close()
    return {'what': 'C2 planner (4096 envs, 50 sims, device epilogue -> HBM replay) and the HIP learner (batch %d gathered from the replay) taking turns on one GPU, event order' % batch,
            'seconds': dt, 'iterations': it, 'env_steps_per_sec': it * moves * B / dt, 'sims_per_sec': it * moves * B * 50 / dt, 'updates_per_sec': it * moves / dt,
            'samples_per_sec': it * moves * batch / dt, 'samples_per_env_step': batch / B, 'last_loss': loss, 'replay_items': int(rp.size),
            'sampling': 'replay.DeviceSampler (Philox indices drawn on the GPU from the device-owned counter; no host synchronisation per update)',
            'replay_flush': 'make_classic_config: acc_seq_length 9999 (items at episode ends)'}


def allreduce_leg(rank, local_rank, world, torch, dist, backend, nbytes=30_400_000, iters=10):
    """The learner-side collective the north-star names ("RCCL over xGMI only for replay / gradient all-reduce"): ONE flat all-reduce of a
    gradient bucket the size of the Gomoku conv net (30.4 MB of fp32), as learner.allreduce_gradients issues it.  N > 1 only."""
    if world < 2:
        return None
    dev = torch.device('cuda', local_rank) if backend == 'nccl' else torch.device('cpu')
    t = torch.ones(nbytes // 4, dtype=torch.float32, device=dev)
    for _ in range(3):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.fill_(1.0)
    if dev.type == 'cuda':
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if dev.type == 'cuda':
        torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters
    ok = bool(abs(float(t[0].item()) - float(world) ** iters) <= 1e-3 * float(world) ** iters)
    tm = torch.tensor([ms], dtype=torch.float64, device=dev)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    ms = float(tm.item())
    alg = nbytes / (ms * 1e-3) / 1e9
    return {'bytes': nbytes, 'ms': ms, 'algbw_GBps': alg, 'busbw_GBps': alg * 2 * (world - 1) / world, 'sum_correct': ok, 'backend': dist.get_backend(),
            'world_size': world, 'what': 'one flat fp32 all-reduce (SUM) of a Gomoku-net-sized gradient bucket, MAX over ranks of the mean time'}


def profiled_traffic(workload, kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this very command
    (profiles/<round>/<workload>/pmc_summary.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be read from inside the run, so this is the profiled
    figure for the default workload sizes -- and only if the summary was measured on the kernel sources this run was
    built from (`_source_fingerprint` == muzero_amd.build.source_fingerprint()): a stale profile yields None, never a
    number that silently belongs to another kernel."""
    from muzero_amd import build as mz_build

    rel = os.path.join('profiles', PROFILE_ROUND, workload, 'pmc_summary.json')
    try:
        doc = json.load(open(os.path.join(REPO, rel)))
        pm = doc['pmc']
    except (OSError, ValueError, KeyError):
        return None, f'{rel}: not found'
    fp = mz_build.source_fingerprint()
    if doc.get('_source_fingerprint') != fp:
        return None, f'{rel} was measured on kernel sources {doc.get("_source_fingerprint")} (commit {doc.get("_git_head")}); this run is built from {fp}: stale, not reported'
    for name, e in pm.items():
        if kernel_substr in name and '_hbm_bytes_per_launch' in e:
            return e['_hbm_bytes_per_launch']['total'], (f'{rel} ({name}; rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per launch; kernel sources {fp}, '
                                                         f'commit {doc.get("_git_head")})')
    return None, f'{rel}: kernel not in the summary'


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without torchrun: start N rank processes (one per GPU) and relay rank 0's JSON line.
    Runs BEFORE anything touches HIP in this process (the parent never initialises a GPU and never exec()s); the
    library is built once here so that the ranks only load it.  Mirrors the reference's topology of N actor processes
    (classic/run_training.py:168-186)."""
    import socket
    import subprocess

    from muzero_amd import build as mz_build

    mz_build.build()
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    # rank 0's stdout may carry backend chatter (gloo prints its connection banner there): relay the JSON line only
    for line in out.decode().splitlines():
        if line.startswith('{') and '"metric"' in line:
            sys.stdout.write(line + '\n')
    sys.stdout.flush()
    if any(codes):
        sys.stderr.write(f'bench.py: rank exit codes {codes}\n')
        return 1
    return 0


def sustained_leg(p, T, ms_per_move_hint, flop_per_launch, seconds=2.0):
    """>= `seconds` of back-to-back lock-step moves after the timed region: the clock the chip HOLDS under this load, not
    a 30 ms burst.  Own HIP-event timing; does not change steps / ms_per_step."""
    n = max(20, int(seconds * 1e3 / max(ms_per_move_hint, 1e-3)) + 1)
    p.synchronize()
    p.profile_begin()
    t0 = time.perf_counter()
    p.selfplay_step(T, n)
    p.synchronize()
    wall = time.perf_counter() - t0
    prof = p.profile_end()
    k_ms = prof['search_kernel_ms'] / max(1, prof['search_kernel_launches'])
    ach = flop_per_launch / (k_ms * 1e-3) / 1e12
    return {'moves': n, 'seconds': wall, 'ms_per_move': 1e3 * wall / n, 'avg_launch_ms': k_ms, 'achieved': ach,
            'frac': ach / PEAK_FP32_MFMA_TFLOPS}


def e2e_leg(p, pl, env_kind, T, obs_shape, classic_cfg, moves=300, warm_moves=None):
    """Env steps that END UP IN THE REPLAY as (Transition, priority) items: the planner with the device epilogue attached
    (n-step / MC targets, priorities, K-step unroll windows written into an HBM replay ring on the GPU, pipeline.py:118-165 +
    replay.py:67-75), this GPU only.  Warm-up runs past the first mid-episode flush (acc_seq_length + unroll + td moves)."""
    import torch
    from muzero_amd.replay import PrioritizedReplay

    rp = PrioritizedReplay(1 << 20, 0.0, 0.0, np.random.RandomState(0), device='cuda')
    p.attach_replay(rp, classic_cfg, obs_shape=obs_shape)
    p.selfplay_reset(env_kind)
    warm = 0 if classic_cfg.is_board_game else classic_cfg.acc_seq_length + classic_cfg.unroll_steps + classic_cfg.td_steps + 5
    if warm_moves is not None:
        warm = warm_moves
    p.selfplay_step(T, max(warm, 20))
    p.synchronize()
    n0 = rp.num_added
    t0 = time.perf_counter()
    p.selfplay_step(T, moves)
    p.synchronize()
    dt = time.perf_counter() - t0
    n1 = rp.num_added
    p.detach_replay()
    del rp
    torch.cuda.empty_cache()
    return {'moves': moves, 'seconds': dt, 'env_steps_per_sec': p.B * moves / dt, 'items_into_replay_per_sec': (n1 - n0) / dt,
            'items_added': int(n1 - n0), 'what': 'device epilogue attached: targets, priorities and unroll windows built on the GPU and written '
                                                 'into an HBM replay ring; the host only reads the counter'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--preheat', type=int, default=150, help='untimed setup moves before the warm-up (c2 / c3): ~0.1 s of load takes the GPU out of its idle clocks')
    ap.add_argument('--envs', type=int, default=0, help='environments per GPU (default: that of the workload)')
    ap.add_argument('--sims', type=int, default=0, help='simulations per move (default: that of the workload)')
    ap.add_argument('--workload', default='c2', choices=['c2', 'c3', 'c4', 'c5', 'lunar'],
                    help='c2 (default, the headline line): CartPole MLP; c3: TicTacToe MLP; c4 / c5: the conv-tower configs of BASELINE.json (extra measurements)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-sustained', action='store_true', help='skip the >= 2 s sustained leg')
    ap.add_argument('--no-e2e', action='store_true', help='skip the env-steps-into-replay leg')
    ap.add_argument('--no-configs', action='store_true', help='skip the C3 / C4 / C5 / LunarLander-shaped legs of the default run')
    ap.add_argument('--no-learner', action='store_true', help='skip the learner-step leg (and, with N > 1 ranks, the gradient all-reduce leg)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.stderr.write(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for the wrong GPU count\n')
        return 2

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # MZ_BENCH_BACKEND=gloo: control-flow check of the N > 1 path on a box with fewer GPUs than ranks (ranks then share
    # devices and the barrier / MAX run on CPU tensors); the driver's multi-GPU runs use the default, RCCL.
    backend = os.environ.get('MZ_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank = local_rank % max(1, torch.cuda.device_count())
    red_dev = 'cuda' if backend == 'nccl' else 'cpu'
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(local_rank)

    from muzero_amd import build as mz_build

    if rank == 0:
        mz_build.build()  # locked + atomic rename (muzero_amd/build.py); the other ranks only load the finished file
    if world > 1:
        dist.barrier()

    if args.workload == 'lunar':  # the LunarLander-shaped leg alone (diagnostics; part of `configs` in the default run)
        rec = lunar_leg(rank, local_rank, world, torch, dist, red_dev, steps=args.steps, warm=args.warmup, preheat=args.preheat)
        if rank == 0:
            print(json.dumps(rec), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 0
    if args.workload != 'c2':
        return run_conv_workload(args, args.workload, rank, local_rank, world, torch, dist, red_dev, backend)

    from helpers import build_mlp, mlp_case
    from muzero_amd import planner as pl

    net = build_mlp(mlp_case('cartpole'))  # 512/64/31, seeded random init
    B, S = args.envs or 4096, args.sims or 50
    cfg = pl.make_mz_config(net.planner_spec(), None, num_envs=B, seed=1000 + rank, num_simulations=S, discount=0.997,
                            root_dirichlet_alpha=0.25, root_exploration_eps=0.25)
    p = pl.Planner(cfg, local_rank)
    p.load_state_dict(net.state_dict())
    p.selfplay_reset(pl.ENV_CARTPOLE)

    def sync():
        p.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # Setup, before the contract's warm-up: bring the GPU out of its idle clocks.  The first ~15 ms of kernels after an idle
    # period run ~7 % slower (tools/idle_clock_probe.py: 0.807 ms per move for moves 1-20 after start-up or after 3 s of
    # idleness, 0.752 from move 21 on, 0.755 straight after an env reset on a busy GPU -- it is the clock, not the episode
    # phase), and W = 5 warm-up moves are 4 ms.  Untimed, disclosed as config.preheat_moves.
    if args.preheat:
        p.selfplay_step(1.0, args.preheat)
    p.selfplay_step(1.0, args.warmup)
    sync()
    p.profile_begin()
    t0 = time.perf_counter()
    p.selfplay_step(1.0, args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    prof = p.profile_end()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    counters = p.selfplay_counters()
    dispatch = p.describe()  # the instantiation the timed launches ran, as the library reports it (mz_planner_describe)
    flop_per_launch = B * (S * FLOP_PER_SIM + FLOP_PER_ROOT)
    sustained = None if args.no_sustained else sustained_leg(p, 1.0, 1e3 * elapsed / args.steps, flop_per_launch)
    e2e = None
    if not args.no_e2e:
        import types

        # the reference's classic-control settings (make_classic_config, config.py:170-201): acc_seq_length 9999 -- items leave an env only at
        # its episode ends (pipeline.py:118-165) --, unroll 5, td_steps 10; warm-up past the first episode ends
        e2e = e2e_leg(p, pl, pl.ENV_CARTPOLE, 1.0, (4, 5),
                      types.SimpleNamespace(is_board_game=False, acc_seq_length=9999, unroll_steps=5, td_steps=10, discount=0.997), warm_moves=120)
        e2e['settings'] = 'make_classic_config: acc_seq_length 9999 (flush at episode ends only), unroll 5, td_steps 10'
        e2e['fraction_of_planner_rate'] = e2e['env_steps_per_sec'] / (B * args.steps / elapsed)
        # the Atari-style mid-episode flush (make_atari_config's acc_seq_length 200, config.py:203-232) on the same planner: round 4's `e2e` figure
        fl = e2e_leg(p, pl, pl.ENV_CARTPOLE, 1.0, (4, 5),
                     types.SimpleNamespace(is_board_game=False, acc_seq_length=200, unroll_steps=5, td_steps=10, discount=0.997))
        fl['settings'] = 'Atari-style flush: acc_seq_length 200 (make_atari_config), unroll 5, td_steps 10'
        fl['fraction_of_planner_rate'] = fl['env_steps_per_sec'] / (B * args.steps / elapsed)
        e2e['atari_style_flush'] = fl
    p.close()
    del p
    torch.cuda.empty_cache()
    configs = None if args.no_configs or args.envs or args.sims else configs_leg(args, rank, local_rank, world, torch, dist, red_dev)
    if configs is not None and 'lunar' in os.environ.get('MZ_BENCH_CONFIGS', 'lunar').split(','):
        lunar = lunar_leg(rank, local_rank, world, torch, dist, red_dev)
        if rank == 0:
            configs['lunar'] = lunar
    learner_rec = None if args.no_learner else learner_leg(rank, local_rank, world, torch, dist, backend)
    train_rec = None if args.no_learner else train_leg(rank, local_rank, world, torch, dist)
    allreduce_rec = None if args.no_learner else allreduce_leg(rank, local_rank, world, torch, dist, backend)
    # every rank's own rate (the line's `value` is the job's: all ranks' simulations over the slowest rank's time)
    my_rate = B * S * args.steps / (prof['search_kernel_ms'] * 1e-3) if prof['search_kernel_ms'] > 0 else 0.0
    per_rank = [my_rate]
    if world > 1:
        tr = torch.zeros(world, dtype=torch.float64, device=red_dev)
        tr[rank] = my_rate
        dist.all_reduce(tr, op=dist.ReduceOp.SUM)
        per_rank = [float(x) for x in tr.tolist()]

    if rank == 0:
        total_sims = world * B * S * args.steps
        sims_per_s = total_sims / elapsed
        k_ms = prof['search_kernel_ms'] / max(1, prof['search_kernel_launches'])
        achieved = flop_per_launch / (k_ms * 1e-3) / 1e12
        # HBM bytes per launch of the same kernel from the committed rocprofv3 PMC passes (separate --pmc runs of this very
        # command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Only valid for the profiled workload.
        traffic, traffic_src = (profiled_traffic('c2', 'k_search_fast<512') if (B == 4096 and S == 50) else (None, 'non-default workload size'))
        out = {
            'metric': 'self-play MCTS sims/sec (env-steps/sec = value / sims_per_move)',
            'value': sims_per_s,
            'unit': 'sims/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {
                'workload': 'C2: CartPole-v1 self-play, 50 sims/move, 4096 parallel envs per MI355X, MuZeroMLPNet 512/64/31, batched tree',
                'envs_per_gpu': B, 'sims_per_move': S, 'num_actions': 2, 'parallelism': f'env-sharded x{world}',
                'randomness': 'on-device Philox', 'weights': 'seeded random init', 'preheat_moves': args.preheat,
            },
            'env_steps_per_sec': sims_per_s / S,
            'episodes_finished_rank0': counters['episodes'],
            'roofline': {
                'bound': 'mfma', 'kernel': dispatch, 'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / PEAK_FP32_MFMA_TFLOPS,
                # the same FLOPs over the driver-timed step (env kernels and launch gaps included), not only the search kernel
                'frac_step': flop_per_launch / (1e-3 * 1e3 * elapsed / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                'traffic': traffic, 'traffic_unit': 'bytes of HBM per launch', 'traffic_source': traffic_src,
                'algorithmic_hbm_bytes_per_launch': B * (S * 2 * 64 * 4 + 20 * 4 + 64 * 4 + 2 * 8 + 16),
                'avg_launch_ms': k_ms, 'launches': prof['search_kernel_launches'], 'flop_per_launch': flop_per_launch,
                'flop_per_sim': FLOP_PER_SIM, 'timed_with': 'hipEvent pairs on the planner stream',
            },
            'sustained': sustained,
            'e2e': e2e,
            'configs': configs,
            'learner': learner_rec,
            'train': train_rec,
            'allreduce': allreduce_rec,
            'per_rank': {'sims_per_sec_by_kernel_time': per_rank, 'min': min(per_rank), 'max': max(per_rank), 'sum': sum(per_rank)},
            'distributed': dist_record(dist, world, backend),
        }
        if not args.no_cpu_baseline:  # rank 0's host cores; under N > 1 a short sample (the other ranks wait at the final barrier)
            out['cpu_baseline'] = cpu_baseline(S, sample_envs=B, budget_s=20.0 if world == 1 else 3.0)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    sys.exit(main() or 0)
