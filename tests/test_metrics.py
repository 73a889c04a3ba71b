"""f3: the reference's metric names (trackers.py:74-80, 113-117, 165-189) as JSON lines."""
import json
import types

import numpy as np

from muzero_amd import metrics as mzm


def _tags(path):
    return [json.loads(l)['tag'] for l in open(path)]


def test_actor_metrics_batch_equals_reference_tracker_semantics(tmp_path):
    """Two envs in lock-step: every finished episode logs the four actor tags with its own return / length, against the
    number of env steps played so far."""
    path = str(tmp_path / 'run' / 'actor0.jsonl')
    t = mzm.ActorMetrics(path, 2)
    reward = np.array([[1, 1], [1, 2], [1, 1], [5, 1]], np.float32)
    done = np.array([[0, 0], [0, 1], [1, 0], [0, 0]], np.uint8)
    t.moves(reward[:2], done[:2])
    t.moves(reward[2:], done[2:])
    t.close()
    assert t.num_episodes == 2
    assert set(_tags(path)) == {'actor(env_steps)/num_episodes', 'actor(env_steps)/episode_return', 'actor(env_steps)/episode_steps',
                                'actor(env_steps)/step_rate(second)'}
    steps, ret = mzm.read_scalars(path, 'actor(env_steps)/episode_return')
    np.testing.assert_array_equal(ret, [3.0, 3.0])      # env 1: 1 + 2 after two moves; env 0: 1 + 1 + 1 after three
    np.testing.assert_array_equal(steps, [4, 5])        # 2 envs x 1 move + env index 1 + 1; 2 x 2 moves + 0 + 1
    _, length = mzm.read_scalars(path, 'actor(env_steps)/episode_steps')
    np.testing.assert_array_equal(length, [2, 3])
    _, n = mzm.read_scalars(path, 'actor(env_steps)/num_episodes')
    np.testing.assert_array_equal(n, [1, 2])
    _, rate = mzm.read_scalars(path, 'actor(env_steps)/step_rate(second)')
    assert (rate > 0).all()


def test_learner_and_evaluator_tags(tmp_path):
    cfg = types.SimpleNamespace(metrics_dir=str(tmp_path), use_tensorboard=False)
    lp = mzm.run_file(cfg, 'learner', 'exp1')
    assert lp.endswith('exp1_learner.jsonl')
    m = mzm.LearnerMetrics(lp)
    m.step(2.5, 1e-3, 1)
    m.step(2.0, 1e-3, 2)
    m.close()
    assert _tags(lp)[:3] == ['learner(train_steps)/loss', 'learner(train_steps)/learning_rate', 'learner(train_steps)/step_rate(minutes)']
    steps, loss = mzm.read_scalars(lp, 'learner(train_steps)/loss')
    np.testing.assert_array_equal(steps, [1, 2])
    np.testing.assert_array_equal(loss, [2.5, 2.0])
    ep = mzm.run_file(cfg, 'evaluator')
    e = mzm.EvaluatorMetrics(ep)
    e.step([10.0, 20.0], [10, 20], 100)
    e.board_game_step(-1990.0, 7, 200)
    e.close()
    assert _tags(ep) == ['evaluator(train_steps)/mean_episode_return', 'evaluator(train_steps)/mean_episode_steps',
                         'evaluator(train_steps)/elo_rating', 'evaluator(train_steps)/episode_steps']
    assert mzm.read_scalars(ep, 'evaluator(train_steps)/mean_episode_return')[1][0] == 15.0
    # nothing is written unless asked for (use_tensorboard False, no metrics_dir): the reference's default for tests
    assert mzm.run_file(types.SimpleNamespace(use_tensorboard=False), 'learner') is None
    assert mzm.run_file(types.SimpleNamespace(use_tensorboard=True), 'learner') == 'runs/learner.jsonl'


def test_run_training_writes_learner_metrics(tmp_path):
    import queue
    import threading
    import torch
    from helpers import build_mlp, mlp_case
    from muzero_amd import learner
    from muzero_amd.config import make_tictactoe_config
    from muzero_amd.replay import PrioritizedReplay, Transition

    net, actor = build_mlp(mlp_case('tiny_mse')), build_mlp(mlp_case('tiny_mse'))
    net.train()
    cfg = make_tictactoe_config(num_training_steps=4, batch_size=4, min_replay_size=8, use_tensorboard=False)
    cfg.checkpoint_interval, cfg.train_delay, cfg.metrics_dir = 2, 0.0, str(tmp_path / 'm')
    rs = np.random.RandomState(0)
    rp = PrioritizedReplay(64, 0.0, 0.0, np.random.RandomState(1))
    for _ in range(12):
        rp.add(Transition(rs.uniform(-1, 1, (2, 2, 2)).astype(np.float32), rs.randint(0, 5, 5).astype(np.int8),
                          rs.dirichlet(np.ones(5), 5).astype(np.float32), rs.uniform(-1, 1, 5).astype(np.float32),
                          rs.uniform(-1, 1, 5).astype(np.float32)), 1.0)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[3], gamma=0.1)
    learner.run_training(cfg, net, opt, sched, torch.device('cpu'), actor, rp, queue.Queue(), types.SimpleNamespace(value=0), None, [],
                         threading.Event(), tag='t', stop_grace_seconds=0.0)
    steps, lr = mzm.read_scalars(str(tmp_path / 'm' / 't_learner.jsonl'), 'learner(train_steps)/learning_rate')
    np.testing.assert_array_equal(steps, [1, 2, 3, 4])
    np.testing.assert_allclose(lr, [1e-3, 1e-3, 1e-4, 1e-4])


def test_bench_reports_profiled_traffic_only_for_matching_kernel_sources(monkeypatch):
    """roofline.traffic comes from a committed rocprofv3 PMC summary; it must belong to the kernel sources the run is built
    from.  bench.profiled_traffic compares the summary's `_source_fingerprint` with muzero_amd.build.source_fingerprint() and returns
    None -- with the reason -- for a stale profile."""
    import json
    import os
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import bench
    from muzero_amd import build as mz_build

    path = os.path.join(repo, 'profiles', bench.PROFILE_ROUND, 'c2', 'pmc_summary.json')
    doc = json.load(open(path))
    assert doc.get('_source_fingerprint') and doc.get('_git_head'), 'committed summaries carry the build they were measured on'
    monkeypatch.setattr(mz_build, 'source_fingerprint', lambda: doc['_source_fingerprint'])
    traffic, src = bench.profiled_traffic('c2', 'k_search_fast<512')
    assert traffic and traffic > 5e7 and doc['_source_fingerprint'] in src
    monkeypatch.setattr(mz_build, 'source_fingerprint', lambda: 'ffffffffffffffff')
    traffic, src = bench.profiled_traffic('c2', 'k_search_fast<512')
    assert traffic is None and 'stale' in src
