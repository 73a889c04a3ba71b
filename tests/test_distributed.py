"""The N>1 path on CPU: world_size-2 gloo processes shard the environments and aggregate throughput exactly as bench.py
does on GPUs over RCCL (no data-path collective: only a SUM of units and a MAX of elapsed time)."""
import os
import socket
import sys

import torch.multiprocessing as mp

from helpers import REPO


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from muzero_amd import pipeline

    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = pipeline.shard_range(8191, rank, world)
    # rank r "plays" (hi - lo) envs x 50 sims x 3 moves in (1 + r) seconds
    rate, units, secs = pipeline.aggregate_throughput((hi - lo) * 50 * 3, 1.0 + rank)
    dist.barrier()
    q.put((rank, lo, hi, rate, units, secs, pipeline.rank_env()))
    dist.destroy_process_group()


def test_two_rank_sharding_and_aggregation():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, rate0, units0, secs0, env0), (r1, lo1, hi1, rate1, units1, secs1, env1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 4096, 4096, 8191)  # disjoint, contiguous, sizes differ by at most one
    assert units0 == units1 == 8191 * 150 and secs0 == secs1 == 2.0  # SUM of units, MAX of time
    assert rate0 == rate1 == 8191 * 150 / 2.0
    assert env0 == (0, 0, 2) and env1 == (1, 1, 2)


def test_shard_range_covers_everything():
    from muzero_amd.pipeline import shard_range

    for total in (1, 7, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _learner_worker(rank, world, port, out):
    """Data-parallel learner: each rank computes gradients on its own batch; after
    learner.allreduce_gradients every rank holds the mean gradient, and a train_step keeps the weights identical."""
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import types

    import numpy as np
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)

    from helpers import build_mlp, mlp_case
    from muzero_amd import learner
    from muzero_amd.replay import Transition

    torch.set_num_threads(1)
    net = build_mlp(mlp_case('tiny'))
    net.train()
    rs = np.random.RandomState(100 + rank)  # different data per rank
    B, K, A = 4, 5, 3
    tr = Transition(rs.uniform(-1, 1, (B, 3, 4)).astype(np.float32), rs.randint(0, A, (B, K)).astype(np.int8),
                    rs.dirichlet(np.ones(A), (B, K)).astype(np.float32), rs.uniform(-2, 2, (B, K)).astype(np.float32),
                    rs.uniform(-1, 1, (B, K)).astype(np.float32))
    w = torch.ones(B)
    loss, _ = learner.calc_loss(net, torch.device('cpu'), tr, w)
    loss.backward()
    local = [p.grad.clone() for p in net.parameters()]
    learner.allreduce_gradients(net, bucket_bytes=4096)  # several buckets
    reduced = [p.grad.clone() for p in net.parameters()]
    gathered = [None] * world
    dist.all_gather_object(gathered, [g.numpy() for g in local])
    mean = [np.mean([gathered[r][i] for r in range(world)], axis=0) for i in range(len(local))]
    ok_mean = all(np.allclose(reduced[i].numpy(), mean[i], rtol=1e-6, atol=1e-8) for i in range(len(local)))
    # one full update on different batches: weights stay in lock-step across ranks
    cfg = types.SimpleNamespace(clip_grad=True, max_grad_norm=5.0)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[10], gamma=0.1)
    learner.train_step(cfg, net, opt, sched, torch.device('cpu'), tr, w)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    both = [None] * world
    dist.all_gather_object(both, flat.numpy())
    if rank == 0:
        out.put(dict(ok_mean=ok_mean, same=bool(np.array_equal(both[0], both[1])), differ_local=not np.allclose(gathered[0][0], gathered[1][0])))
    dist.barrier()
    dist.destroy_process_group()


def test_learner_gradient_allreduce_world2():
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_learner_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    res = out.get()
    assert res['differ_local'] and res['ok_mean'] and res['same']


def _actor_side(actor, counter, ready, go1, mid, go2, out):
    """Runs in a SPAWNED process, like the reference's actors (classic/run_training.py:168-186): watches the shared network."""
    sys.path.insert(0, REPO)
    import types

    from muzero_amd import pipeline

    cfg = types.SimpleNamespace(checkpoint_interval=5)
    k0 = pipeline.weights_key(actor, counter, cfg)
    w0 = float(next(actor.parameters()).flatten()[0])
    ready.set()
    go1.wait(60)  # the train-step counter has crossed the boundary, the weights have NOT been published yet
    k1 = pipeline.weights_key(actor, counter, cfg)
    mid.set()
    go2.wait(60)  # now they have
    k2 = pipeline.weights_key(actor, counter, cfg)
    w2 = float(next(actor.parameters()).flatten()[0])
    out.put((k0 == k1, k1 != k2, w0, w2))


def test_weight_refresh_signal_crosses_processes():
    """ADVICE r1 + r2: tensor._version does not cross processes, and the train-step counter crosses a checkpoint boundary
    BEFORE the learner has copied the weights (learner.run_training: counter += 1, checkpoint, barrier, load_state_dict).  An
    actor in another process must not take the counter for the signal (it would reload the old values and stay one
    checkpoint behind); it must see the shared `weights_epoch` buffer, bumped after the new values are in place."""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import torch

    from helpers import build_mlp, mlp_case

    ctx = mp.get_context('spawn')
    actor = build_mlp(mlp_case('tiny'))
    actor.share_memory()
    assert actor.weights_epoch.is_shared() and 'weights_epoch' not in actor.state_dict()  # checkpoint layout unchanged
    counter = ctx.Value('i', 3)
    ready, go1, mid, go2, out = ctx.Event(), ctx.Event(), ctx.Event(), ctx.Event(), ctx.SimpleQueue()
    proc = ctx.Process(target=_actor_side, args=(actor, counter, ready, go1, mid, go2, out))
    proc.start()
    assert ready.wait(120)
    counter.value = 5  # run_training: train_steps_counter.value += 1 happens first ...
    go1.set()
    assert mid.wait(60)
    new = {k: v + 1.0 if v.dtype.is_floating_point else v for k, v in actor.state_dict().items()}
    actor.publish_weights(new)  # ... the copy (in place: the shared storage changes) and the signal after it
    go2.set()
    proc.join(timeout=60)
    assert proc.exitcode == 0
    quiet_before, fired_after, w0, w2 = out.get()
    assert quiet_before and fired_after and abs((w2 - w0) - 1.0) < 1e-6


def _train_worker(rank, world, port, out, tmp):
    """run_training on two learner ranks whose replay shards warm up at different times and of which only one sees the stop
    event: all decisions around the gradient all-reduce are collective, nobody hangs, rank 0 alone writes checkpoints."""
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import queue
    import threading
    import time
    import types

    import numpy as np
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from helpers import build_mlp, mlp_case
    from muzero_amd import learner
    from muzero_amd.config import make_tictactoe_config
    from muzero_amd.replay import PrioritizedReplay, Transition

    torch.set_num_threads(1)
    net, actor = build_mlp(mlp_case('tiny_mse')), build_mlp(mlp_case('tiny_mse'))
    net.train()
    cfg = make_tictactoe_config(num_training_steps=1000, batch_size=4, min_replay_size=8, use_tensorboard=False)
    cfg.checkpoint_interval, cfg.train_delay = 3, 0.0
    rs = np.random.RandomState(rank)
    rp = PrioritizedReplay(64, 0.0, 0.0, np.random.RandomState(1 + rank))

    def fill():
        if rank == 1:
            time.sleep(1.0)  # this shard warms up later: rank 0 must wait for it instead of entering the all-reduce alone
        for _ in range(12):
            rp.add(Transition(rs.uniform(-1, 1, (2, 2, 2)).astype(np.float32), rs.randint(0, 5, 5).astype(np.int8),
                              rs.dirichlet(np.ones(5), 5).astype(np.float32), rs.uniform(-1, 1, 5).astype(np.float32),
                              rs.uniform(-1, 1, 5).astype(np.float32)), 1.0)

    threading.Thread(target=fill).start()
    stop = threading.Event()
    counter = types.SimpleNamespace(value=0)

    class StopAfter:  # only rank 1 ever raises the stop flag, after 7 of its own steps
        def is_set(self):
            return stop.is_set() or (rank == 1 and counter.value >= 7)

        def set(self):
            stop.set()

    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[100], gamma=0.1)
    files = []
    # a stop raised mid-run ends the loop only through num_training_steps in the reference; here rank 1 lowers its own limit
    if rank == 1:
        cfg.num_training_steps = 7
    learner.run_training(cfg, net, opt, sched, torch.device('cpu'), actor, rp, queue.Queue(), counter, tmp, files, StopAfter(), stop_grace_seconds=0.0)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    both = [None] * world
    dist.all_gather_object(both, (flat, counter.value, sorted(os.listdir(tmp))))
    if rank == 0:
        out.put(dict(same=bool(np.array_equal(both[0][0], both[1][0])), steps=(both[0][1], both[1][1]), files=both[0][2]))
    dist.barrier()
    dist.destroy_process_group()


def test_run_training_world2_is_collective(tmp_path):
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, out, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    res = out.get()
    assert res['same'] and res['steps'] == (7, 7)  # both ranks stopped together when rank 1 reached its limit
    assert res['files'] == ['train_steps_3', 'train_steps_6', 'train_steps_7_final']  # written once, by rank 0


def test_bench_gpus_flag_never_reports_a_single_gpu_line():
    """`python bench.py --gpus N` without torchrun must start N ranks itself, and a launch whose WORLD_SIZE
    disagrees with --gpus must refuse to print a line.  Without a GPU the ranks fail (no CPU fallback) -- the point here is
    that no run asked for 2 GPUs can ever print `"n_gpus": 1`."""
    import subprocess

    env = dict(os.environ, MZ_BENCH_BACKEND='gloo')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu-baseline',
                        '--no-sustained', '--no-e2e'], capture_output=True, text=True, timeout=600, env=env)
    assert '"n_gpus": 1' not in r.stdout
    import torch

    if not torch.cuda.is_available():
        assert r.returncode != 0  # the ranks cannot create a planner without a GPU, and the launcher says so
    r2 = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], capture_output=True,
                        text=True, timeout=120, env=dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'))
    assert r2.returncode == 2 and 'refusing' in r2.stderr and r2.stdout.strip() == ''


def _worker8(rank, world, port, q):
    """World-size-8 rehearsal: uneven env shards, SUM / MAX aggregation, and the learner's bucketed gradient
    all-reduce over eight ranks -- the plumbing the first 8-GPU lease will run over RCCL."""
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import numpy as np
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from helpers import build_mlp, mlp_case
    from muzero_amd import learner, pipeline

    torch.set_num_threads(1)
    total = 4099  # not divisible by 8: three ranks carry one env more
    lo, hi = pipeline.shard_range(total, rank, world)
    rate, units, secs = pipeline.aggregate_throughput((hi - lo) * 50 * 2, 1.0 + 0.125 * rank)
    net = build_mlp(mlp_case('tiny'))  # same seed on every rank: identical weights
    for i, p in enumerate(net.parameters()):
        p.grad = torch.full_like(p, float(rank + 1)) * (i + 1)
    learner.allreduce_gradients(net, bucket_bytes=2048)  # several buckets
    want = np.mean([r + 1 for r in range(world)])
    ok = all(torch.allclose(p.grad, torch.full_like(p, float(want * (i + 1)))) for i, p in enumerate(net.parameters()))
    dist.barrier()
    q.put((rank, lo, hi, rate, units, secs, ok, pipeline.rank_env()))
    dist.destroy_process_group()


def test_eight_rank_sharding_aggregation_and_allreduce():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(8))
    spans = [(r[1], r[2]) for r in res]
    assert spans[0][0] == 0 and spans[-1][1] == 4099 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sizes = [hi - lo for lo, hi in spans]
    assert sorted(sizes) == [512] * 5 + [513] * 3
    for r in res:
        assert r[4] == 4099 * 100 and r[5] == 1.875 and r[3] == 4099 * 100 / 1.875  # SUM of units, MAX of time, on every rank
        assert r[6] and r[7] == (r[0], r[0], 8)
