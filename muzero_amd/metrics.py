"""Run metrics under the reference's tag names (muzero/trackers.py:74-80, 113-117, 165-189), written as JSON lines instead
of tensorboard event files: one object {"tag", "value", "step", "wall"} per scalar, so learning curves can be compared tag
for tag with the reference's screenshots (`actor(env_steps)/step_rate(second)` is the throughput BASELINE.md quotes).

A run writes `<metrics_dir>/<tag_>actor<rank>.jsonl`, `<tag_>learner.jsonl`, `<tag_>evaluator.jsonl`; `metrics_dir` is
`config.metrics_dir` if set, else "runs" when `config.use_tensorboard` is true (where the reference puts its event
files, trackers.py:200-215), else nothing is written."""
import json
import os
import timeit
from typing import List, Optional

import numpy as np


def metrics_dir(config) -> Optional[str]:
    d = getattr(config, 'metrics_dir', None)
    if d:
        return d
    return 'runs' if getattr(config, 'use_tensorboard', False) else None


def run_file(config, role: str, tag: Optional[str] = None) -> Optional[str]:
    d = metrics_dir(config)
    if d is None:
        return None
    return os.path.join(d, f'{tag}_{role}.jsonl' if tag else f'{role}.jsonl')


class ScalarWriter:
    """add_scalar(tag, value, step) like tensorboard's SummaryWriter; path None = discard."""

    def __init__(self, path: Optional[str]):
        self._f = None
        if path:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            self._f = open(path, 'a', buffering=1)
        self._t0 = timeit.default_timer()

    def add_scalar(self, tag: str, value, step: int) -> None:
        if self._f is not None:
            self._f.write(json.dumps({'tag': tag, 'value': float(value), 'step': int(step), 'wall': timeit.default_timer() - self._t0}) + '\n')

    def close(self) -> None:
        if self._f is not None:
            self._f.close()
            self._f = None


def _rate(steps: int, start: float) -> float:
    return steps / (timeit.default_timer() - start) if steps > 0 else float('nan')


class ActorMetrics:
    """trackers.py:29-95 for a lock-step batch of environments: `moves(reward[M, B], done[M, B])` accounts M moves of B envs;
    every finished episode logs num_episodes / episode_return / episode_steps / step_rate(second) against the number of env
    steps played so far (the reference's one-env tracker is the B = 1 case)."""

    def __init__(self, path: Optional[str], num_envs: int):
        self._w = ScalarWriter(path)
        self._ret = np.zeros(num_envs, np.float64)
        self._len = np.zeros(num_envs, np.int64)
        self._steps = 0
        self._episodes = 0
        self._start = timeit.default_timer()

    def moves(self, reward, done) -> None:
        reward, done = np.asarray(reward), np.asarray(done)
        for m in range(reward.shape[0]):
            self._ret += reward[m]
            self._len += 1
            for b in np.nonzero(done[m])[0]:
                # env steps so far: whole moves before this one plus the envs of this move up to b (the reference's
                # `_num_steps_since_reset` counted step by step)
                tb_steps = self._steps + int(b) + 1
                self._episodes += 1
                self._w.add_scalar('actor(env_steps)/num_episodes', self._episodes, tb_steps)
                self._w.add_scalar('actor(env_steps)/episode_return', self._ret[b], tb_steps)
                self._w.add_scalar('actor(env_steps)/episode_steps', self._len[b], tb_steps)
                self._w.add_scalar('actor(env_steps)/step_rate(second)', _rate(tb_steps, self._start), tb_steps)
                self._ret[b] = 0.0
                self._len[b] = 0
            self._steps += reward.shape[1]

    @property
    def num_episodes(self) -> int:
        return self._episodes

    def close(self) -> None:
        self._w.close()


class LearnerMetrics:
    """trackers.py:98-131: loss, learning_rate, step_rate(minutes) per train step."""

    def __init__(self, path: Optional[str]):
        self._w = ScalarWriter(path)
        self._start = timeit.default_timer()

    def step(self, loss, lr, train_steps: int) -> None:
        self._w.add_scalar('learner(train_steps)/loss', loss, train_steps)
        self._w.add_scalar('learner(train_steps)/learning_rate', lr, train_steps)
        self._w.add_scalar('learner(train_steps)/step_rate(minutes)', _rate(train_steps, self._start) * 60, train_steps)

    def close(self) -> None:
        self._w.close()


class EvaluatorMetrics:
    """trackers.py:134-170 (classic / Atari evaluation episodes) and :173-189 (board games: Elo of the newest checkpoint)."""

    def __init__(self, path: Optional[str]):
        self._w = ScalarWriter(path)

    def step(self, episode_returns: List[float], episode_steps: List[int], train_steps: int) -> None:
        self._w.add_scalar('evaluator(train_steps)/mean_episode_return', np.mean(episode_returns), train_steps)
        self._w.add_scalar('evaluator(train_steps)/mean_episode_steps', np.mean(episode_steps), train_steps)

    def board_game_step(self, elo, episode_steps, train_steps: int) -> None:
        self._w.add_scalar('evaluator(train_steps)/elo_rating', elo, train_steps)
        self._w.add_scalar('evaluator(train_steps)/episode_steps', episode_steps, train_steps)

    def close(self) -> None:
        self._w.close()


def read_scalars(path: str, tag: str):
    """(steps, values) of one tag from a metrics file."""
    steps, values = [], []
    with open(path) as f:
        for line in f:
            o = json.loads(line)
            if o['tag'] == tag:
                steps.append(o['step'])
                values.append(o['value'])
    return np.array(steps), np.array(values)
