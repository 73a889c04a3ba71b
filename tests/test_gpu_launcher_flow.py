"""The reference's process topology (classic/run_training.py:59-194) driven through this package's drop-in functions on one GPU: a SPAWNED
actor process (`run_self_play` on a shared-memory `actor_network`, its own planner on the GPU), a data collector thread
(`run_data_collector`: queue -> replay) and the learner (`run_training`, update on the HIP learner kernels) in the launching process."""
import multiprocessing as mp
import os
import sys
import threading

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _actor_main(cfg, actor_network, data_queue, counter, stop_event, out):
    sys.path.insert(0, REPO)
    import torch

    from muzero_amd import pipeline

    try:
        steps = pipeline.run_self_play(cfg, 0, actor_network, torch.device('cuda', 0), 'CartPole-v1', data_queue, counter, stop_event, moves_per_drain=4)
        out.put(('ok', steps))
    except Exception as e:  # noqa: BLE001 -- reported to the parent
        out.put(('error', repr(e)))


def test_spawned_actor_collector_and_hip_learner_train_together(tmp_path):
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import torch

    from helpers import build_mlp, mlp_case
    from muzero_amd import learner
    from muzero_amd.config import make_classic_config
    from muzero_amd.pipeline import load_checkpoint
    from muzero_amd.replay import PrioritizedReplay

    ctx = mp.get_context('spawn')
    dev = torch.device('cuda', 0)
    cfg = make_classic_config(num_training_steps=24, batch_size=32, min_replay_size=96, use_tensorboard=False)
    cfg.num_envs, cfg.num_simulations, cfg.checkpoint_interval, cfg.train_delay = 32, 8, 8, 0.0
    network = build_mlp(mlp_case('cartpole')).to(dev)
    actor_network = build_mlp(mlp_case('cartpole'))
    actor_network.share_memory()  # classic/run_training.py:100
    before = {k: v.clone() for k, v in actor_network.state_dict().items()}
    hl = learner.make_hip_learner(cfg, network, dev)
    replay = PrioritizedReplay(4096, 0.0, 0.0, np.random.RandomState(0), device='cuda')  # (the HIP learner gathers its batch from HBM)
    data_queue, counter, stop_event, out = ctx.SimpleQueue(), ctx.Value('i', 0), ctx.Event(), ctx.SimpleQueue()
    actor = ctx.Process(target=_actor_main, args=(cfg, actor_network, data_queue, counter, stop_event, out))
    actor.start()
    collector = threading.Thread(target=learner.run_data_collector, args=(data_queue, replay))
    collector.start()
    files = []
    learner.run_training(cfg, network, hl.optimizer, hl.lr_scheduler, dev, actor_network, replay, data_queue, counter, str(tmp_path), files, stop_event,
                         stop_grace_seconds=1.0)
    actor.join(timeout=120)
    collector.join(timeout=60)
    assert actor.exitcode == 0 and not collector.is_alive()
    status, played = out.get()
    assert status == 'ok', played
    assert played > 0 and replay.num_added >= cfg.min_replay_size
    assert counter.value == 24 and hl.steps == 24 and len(files) == 3
    # the actor's shared-memory copy holds the weights of the last checkpoint boundary (step 24), which differ from the initial ones
    for k, v in actor_network.state_dict().items():
        assert torch.equal(v, network.state_dict()[k].cpu()), k
    assert any(not torch.equal(before[k], v) for k, v in actor_network.state_dict().items())
    ck = load_checkpoint(str(tmp_path / 'train_steps_24_final'), torch.device('cpu'))
    assert ck['train_steps'] == 24 and set(ck) == {'network', 'optimizer', 'lr_scheduler', 'train_steps'}


def test_threaded_actor_with_device_replay_and_hip_learner(tmp_path):
    """The single-process layout: an actor THREAD whose planner writes finished items straight into the HBM replay (device epilogue,
    `run_self_play(..., data_queue=<PrioritizedReplay on the GPU>)`) while `run_training` draws batches from the same ring and updates on
    the HIP learner kernels; TicTacToe (two players, MC-return targets, int8 board observations)."""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import queue
    import types

    import torch

    from helpers import build_mlp, mlp_case
    from muzero_amd import learner, pipeline
    from muzero_amd.config import make_tictactoe_config
    from muzero_amd.replay import PrioritizedReplay

    dev = torch.device('cuda', 0)
    cfg = make_tictactoe_config(num_training_steps=30, batch_size=32, min_replay_size=128, use_tensorboard=False)
    cfg.num_envs, cfg.num_simulations, cfg.checkpoint_interval, cfg.train_delay = 64, 8, 10, 0.0
    network = build_mlp(mlp_case('tictactoe')).to(dev)
    actor_network = build_mlp(mlp_case('tictactoe'))
    hl = learner.make_hip_learner(cfg, network, dev)
    replay = PrioritizedReplay(8192, 0.0, 0.0, np.random.RandomState(0), device='cuda')
    counter, stop_event, played, errors = types.SimpleNamespace(value=0), threading.Event(), [0], []

    def actor():
        try:
            played[0] = pipeline.run_self_play(cfg, 0, actor_network, dev, 'TicTacToe', replay, counter, stop_event, moves_per_drain=4)
        except Exception as e:  # noqa: BLE001 -- reported by the assertion below
            errors.append(repr(e))
            stop_event.set()

    th = threading.Thread(target=actor)
    th.start()
    files = []
    learner.run_training(cfg, network, hl.optimizer, hl.lr_scheduler, dev, actor_network, replay, queue.SimpleQueue(), counter, str(tmp_path), files,
                         stop_event, stop_grace_seconds=0.5)
    th.join(timeout=120)
    assert not th.is_alive() and not errors, errors
    assert counter.value == 30 and hl.steps == 30 and len(files) == 3
    assert played[0] > 0 and replay.num_added >= cfg.min_replay_size
    for k, v in actor_network.state_dict().items():
        assert torch.equal(v.cpu(), network.state_dict()[k].cpu()), k
    # the replay is usable from the host again after the actor closed its planner (the epilogue detached)
    batch, idx, w = replay.sample(8)
    assert batch.state.shape[0] == 8 and np.isfinite(batch.value).all()


def test_run_training_uses_a_prepared_graph_for_conv_nets(tmp_path):
    """run_training with the one-graph update captured by the launcher (learner.prepare_graphed_step) on a board-game conv net: the loop
    finds the step on the optimizer, checkpoints carry the reference's four entries, the actor network receives the trained weights."""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import queue
    import types

    import torch

    from muzero_amd import learner
    from muzero_amd.config import make_gomoku_config
    from muzero_amd.network import MuZeroBoardGameNet
    from muzero_amd.pipeline import load_checkpoint
    from muzero_amd.replay import PrioritizedReplay, Transition

    dev = torch.device('cuda', 0)
    N = 5
    A, shape = N * N + 1, (9, N, N)
    cfg = make_gomoku_config(num_training_steps=6, batch_size=16, min_replay_size=32, use_tensorboard=False)
    cfg.checkpoint_interval, cfg.train_delay = 3, 0.0
    torch.manual_seed(0)
    net = MuZeroBoardGameNet(shape, A, 1, 16).to(dev)
    actor = MuZeroBoardGameNet(shape, A, 1, 16)
    opt = learner.make_capturable_adam(net, cfg, dev)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[4], gamma=0.1)
    step = learner.prepare_graphed_step(cfg, net, opt, dev, shape, A)
    rs = np.random.RandomState(0)
    K, n = cfg.unroll_steps, 64
    rp = PrioritizedReplay(256, 0.0, 0.0, np.random.RandomState(1), device='cuda')
    rp.add_batch(Transition(rs.randint(0, 2, (n,) + shape).astype(np.float32), rs.randint(0, A, (n, K)).astype(np.int8),
                            rs.dirichlet(np.ones(A), size=(n, K)).astype(np.float32), rs.uniform(-1, 1, (n, K)).astype(np.float32),
                            rs.uniform(-1, 1, (n, K)).astype(np.float32)), np.ones(n))
    before = {k: v.clone() for k, v in net.state_dict().items()}
    counter, stop, files = types.SimpleNamespace(value=0), threading.Event(), []
    learner.run_training(cfg, net, opt, sched, dev, actor, rp, queue.SimpleQueue(), counter, str(tmp_path), files, stop, stop_grace_seconds=0.0)
    assert counter.value == 6 and len(files) == 2 and opt.graphed_step is step
    assert abs(float(opt.param_groups[0]["lr"]) - 0.1 * cfg.lr_init) < 1e-9  # (a float32 device tensor) the schedule crossed its milestone through the device tensor
    assert any(not torch.equal(before[k], v) for k, v in net.state_dict().items() if v.dtype.is_floating_point)
    for k, v in actor.state_dict().items():
        assert torch.equal(v.cpu(), net.state_dict()[k].cpu()), k
    ck = load_checkpoint(str(tmp_path / 'train_steps_6_final'), torch.device('cpu'))
    assert set(ck) == {'network', 'optimizer', 'lr_scheduler', 'train_steps'} and ck['train_steps'] == 6


def test_bench_eight_ranks_over_gloo_on_one_gpu():
    """the N = 8 line of bench.py, end to end, on the one GPU this box has -- eight rank processes (started by bench.py
    itself, as the driver's torchrun would), gloo for the barrier / MAX (`MZ_BENCH_BACKEND=gloo`; the driver's 8-GPU run uses RCCL), every
    rank its own planner and env shard.  The line must say n_gpus 8, carry eight per-rank rates that sum to the value, and scale "weak"."""
    import json
    import os
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MZ_BENCH_BACKEND='gloo')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1', '--preheat', '0', '--no-cpu-baseline',
                        '--no-sustained', '--no-e2e', '--no-configs', '--no-learner'], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['scaling'] == 'weak' and d['steps'] == 3
    rates = d['per_rank']['sims_per_sec_by_kernel_time']
    assert len(rates) == 8 and all(v > 0 for v in rates)
    assert d['value'] > 0 and d['config']['workload'].startswith('C2') and d['config']['parallelism'] == 'env-sharded x8'
    assert d['distributed']['world_size'] == 8
