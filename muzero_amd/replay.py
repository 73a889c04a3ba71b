"""Replay for the learner side of the path -- the reference's `replay.py:27-166` API
(`Transition`, `PrioritizedReplay.add / get / sample / update_priorities / size / num_added / capacity / reset /
get_state / set_state`) over a structure-of-arrays ring buffer instead of a Python list of snappy-compressed tuples.

Storage is one preallocated tensor per `Transition` field (`state [cap, *shape]`, `action int8 [cap, K]`,
`pi_prob f32 [cap, K, A]`, `value f32 [cap, K]`, `reward f32 [cap, K]`) on a torch device: with `device='cuda'` the
ring lives in HBM and `sample_tensors` hands the learner device tensors without a host round trip (a 288 GB MI355X
holds the reference's largest replay, 10^6 Atari transitions of 8x96x96 uint8-equivalent frames, many times over);
with the default `device='cpu'` it is a drop-in for the reference class.  Sampling semantics are the reference's,
draw for draw: uniform replay draws `random_state.uniform(0, size, batch).astype(int64)` (`replay.py:87-89`),
prioritized replay draws from the process-global `np.random.choice` (`replay.py:90-98`; an upstream quirk kept on
purpose) with importance weights `((1/size) / p_i)^beta / max`."""
from typing import Any, List, Mapping, NamedTuple, Optional, Sequence, Text, Tuple

import numpy as np
import torch


class Transition(NamedTuple):
    """replay.py:27-33"""

    state: Optional[np.ndarray]
    action: Optional[np.ndarray]
    pi_prob: Optional[np.ndarray]
    value: Optional[np.ndarray]
    reward: Optional[np.ndarray]


TransitionStructure = Transition(state=None, action=None, pi_prob=None, value=None, reward=None)


class PrioritizedReplay:
    """replay.py:39-142 with array storage.  `priority_exponent == 0` is uniform replay (all launchers' default)."""

    def __init__(self, capacity: int, priority_exponent: float, importance_sampling_exponent: float, random_state: np.random.RandomState,
                 device='cpu'):
        if capacity <= 0:
            raise ValueError(f'Expect capacity to be a positive integer, got {capacity}')
        self.structure = TransitionStructure
        self._capacity = int(capacity)
        self._random_state = random_state
        self._num_added = 0
        self._priority_exponent = priority_exponent
        self._importance_sampling_exponent = importance_sampling_exponent
        self._priorities = np.zeros((capacity,), dtype=np.float32)
        self._device = torch.device(device)
        self._store = None  # dict field -> tensor [capacity, ...], allocated on the first add (shapes come from the data)

    # ---- storage ----
    def _allocate(self, item: Transition) -> None:
        self._store = {}
        for name, x in zip(Transition._fields, item):
            a = np.asarray(x)
            self._store[name] = torch.zeros((self._capacity,) + a.shape, dtype=torch.from_numpy(np.zeros(1, a.dtype)).dtype, device=self._device)

    def add(self, item: Transition, priority: float) -> None:
        """Adds single item to replay (replay.py:67-75)."""
        if not np.isfinite(priority) or priority < 0.0:
            raise ValueError('Priority must be finite and positive.')
        if self._store is None:
            self._allocate(item)
        index = self._num_added % self._capacity
        for name, x in zip(Transition._fields, item):
            self._store[name][index] = torch.from_numpy(np.ascontiguousarray(x))
        self._priorities[index] = priority
        self._num_added += 1

    def add_batch(self, items: Transition, priorities: Sequence[float]) -> None:
        """Adds n items at once (fields stacked on axis 0): the form the device-resident actor produces."""
        pr = np.asarray(priorities, np.float64)
        if not np.isfinite(pr).all() or (pr < 0.0).any():
            raise ValueError('Priority must be finite and positive.')
        n = len(pr)
        if self._store is None:
            self._allocate(Transition(*[np.asarray(x)[0] for x in items]))
        idx = (self._num_added + np.arange(n)) % self._capacity
        tidx = torch.from_numpy(idx).to(self._device)
        for name, x in zip(Transition._fields, items):
            x = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x))
            self._store[name][tidx] = x.to(self._device, dtype=self._store[name].dtype)
        self._priorities[idx] = pr
        self._num_added += n

    def get(self, indices: Sequence[int]) -> List[Transition]:
        """Retrieves items by indices (replay.py:77-79)."""
        return [Transition(*[self._store[f][int(i)].cpu().numpy() for f in Transition._fields]) for i in indices]

    # ---- sampling ----
    def _draw(self, batch_size: int) -> Tuple[np.ndarray, np.ndarray]:
        if self.size < batch_size:
            raise RuntimeError(f'Replay only have {self.size} samples, got sample batch size {batch_size}')
        if self._priority_exponent == 0:
            indices = self._random_state.uniform(0, self.size, size=batch_size).astype(np.int64)
            weights = np.ones_like(indices, dtype=np.float32)
        else:
            priorities = self._priorities[: self.size] ** self._priority_exponent
            probs = priorities / np.sum(priorities)
            indices = np.random.choice(np.arange(probs.shape[0]), size=batch_size, replace=True, p=probs)
            weights = ((1.0 / self.size) / probs[indices]) ** self._importance_sampling_exponent
            weights /= np.max(weights)
        return indices, weights

    def sample_tensors(self, batch_size: int) -> Tuple[Transition, np.ndarray, np.ndarray]:
        """Like `sample`, but the batch stays on the replay's device as torch tensors (no host copy)."""
        indices, weights = self._draw(batch_size)
        tidx = torch.from_numpy(indices).to(self._device)
        return Transition(*[self._store[f].index_select(0, tidx) for f in Transition._fields]), indices, weights

    def sample(self, batch_size: int) -> Tuple[Transition, np.ndarray, np.ndarray]:
        """Samples batch of items from replay, with replacement (replay.py:81-104): numpy arrays stacked on axis 0."""
        batch, indices, weights = self.sample_tensors(batch_size)
        return Transition(*[x.cpu().numpy() for x in batch]), indices, weights

    def update_priorities(self, indices: Sequence[int], priorities: Sequence[float]) -> None:
        """replay.py:106-113"""
        priorities = np.asarray(priorities)
        if not np.isfinite(priorities).all() or (priorities < 0.0).any():
            raise ValueError('Priorities must be finite and positive.')
        for i, p in zip(indices, priorities):
            self._priorities[i] = p

    # ---- bookkeeping (replay.py:115-142) ----
    @property
    def num_added(self) -> int:
        return self._num_added

    @property
    def size(self) -> int:
        return min(self._num_added, self._capacity)

    @property
    def capacity(self) -> int:
        return self._capacity

    def reset(self) -> None:
        self._num_added = 0

    def get_state(self) -> Mapping[Text, Any]:
        return {'num_added': self._num_added, 'storage': None if self._store is None else {k: v.cpu() for k, v in self._store.items()},
                'priorities': self._priorities}

    def set_state(self, state: Mapping[Text, Any]) -> None:
        self._num_added = state['num_added']
        self._store = None if state['storage'] is None else {k: v.to(self._device) for k, v in state['storage'].items()}
        self._priorities = state['priorities']
