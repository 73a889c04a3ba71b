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

    FIELDS = Transition._fields

    def __init__(self, capacity: int, priority_exponent: float, importance_sampling_exponent: float, random_state: np.random.RandomState,
                 device='cpu'):
        if capacity <= 0:
            raise ValueError(f'Expect capacity to be a positive integer, got {capacity}')
        self.structure = TransitionStructure
        self._cap, self._count = int(capacity), 0
        self._rng = random_state
        self._alpha, self._beta = priority_exponent, importance_sampling_exponent
        self._prio = np.zeros(self._cap, dtype=np.float32)
        self._dev = torch.device(device)
        self._ring = None  # field -> tensor [capacity, ...]; allocated on the first add, when the shapes are known

    # ---- storage ----
    @staticmethod
    def _check_priorities(pr) -> None:
        pr = np.asarray(pr, np.float64)
        if not np.isfinite(pr).all() or (pr < 0.0).any():
            raise ValueError('Priority must be finite and positive.')

    def _allocate(self, item: Transition) -> None:
        self._ring = {}
        for name, x in zip(self.FIELDS, item):
            a = np.asarray(x)
            self._ring[name] = torch.zeros((self._cap,) + a.shape, dtype=torch.from_numpy(np.zeros(1, a.dtype)).dtype, device=self._dev)

    def add(self, item: Transition, priority: float) -> None:
        """One item into the ring slot `num_added % capacity` (replay.py:67-75)."""
        self._check_priorities(priority)
        if self._ring is None:
            self._allocate(item)
        slot = self._count % self._cap
        for name, x in zip(self.FIELDS, item):
            self._ring[name][slot] = torch.from_numpy(np.ascontiguousarray(x))
        self._prio[slot] = priority
        self._count += 1

    def add_batch(self, items: Transition, priorities: Sequence[float]) -> None:
        """n items at once (fields stacked on axis 0): the form the device-resident actor produces."""
        self._check_priorities(priorities)
        n = len(priorities)
        if self._ring is None:
            self._allocate(Transition(*[np.asarray(x)[0] for x in items]))
        slots = (self._count + np.arange(n)) % self._cap
        tslots = torch.from_numpy(slots).to(self._dev)
        for name, x in zip(self.FIELDS, items):
            x = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x))
            self._ring[name][tslots] = x.to(self._dev, dtype=self._ring[name].dtype)
        self._prio[slots] = np.asarray(priorities, np.float32)
        self._count += n

    def get(self, indices: Sequence[int]) -> List[Transition]:
        """Items by index (replay.py:77-79)."""
        return [Transition(*[self._ring[f][int(i)].cpu().numpy() for f in self.FIELDS]) for i in indices]

    # ---- sampling ----
    def _draw(self, batch_size: int) -> Tuple[np.ndarray, np.ndarray]:
        n = self.size
        if n < batch_size:
            raise RuntimeError(f'Replay only have {n} samples, got sample batch size {batch_size}')
        if self._alpha == 0:  # uniform: the RandomState handed to the constructor (replay.py:87-89)
            picks = self._rng.uniform(0, n, size=batch_size).astype(np.int64)
            return picks, np.ones_like(picks, dtype=np.float32)
        # proportional: the process-global numpy RNG, as upstream (replay.py:90-98)
        scaled = self._prio[:n] ** self._alpha
        probs = scaled / np.sum(scaled)
        picks = np.random.choice(np.arange(probs.shape[0]), size=batch_size, replace=True, p=probs)
        is_w = ((1.0 / n) / probs[picks]) ** self._beta
        is_w /= np.max(is_w)
        return picks, is_w

    def sample_tensors(self, batch_size: int) -> Tuple[Transition, np.ndarray, np.ndarray]:
        """Like `sample`, but the batch stays on the replay's device as torch tensors (no host copy)."""
        picks, is_w = self._draw(batch_size)
        tpicks = torch.from_numpy(picks).to(self._dev)
        return Transition(*[self._ring[f].index_select(0, tpicks) for f in self.FIELDS]), picks, is_w

    def sample(self, batch_size: int) -> Tuple[Transition, np.ndarray, np.ndarray]:
        """A batch with replacement (replay.py:81-104): numpy arrays stacked on axis 0, the indices, the IS weights."""
        batch, picks, is_w = self.sample_tensors(batch_size)
        return Transition(*[x.cpu().numpy() for x in batch]), picks, is_w

    def update_priorities(self, indices: Sequence[int], priorities: Sequence[float]) -> None:
        """replay.py:106-113 (a repeated index keeps the last value, like the reference's loop)."""
        pr = np.asarray(priorities)
        if not np.isfinite(pr).all() or (pr < 0.0).any():
            raise ValueError('Priorities must be finite and positive.')
        for i, v in zip(indices, pr):
            self._prio[i] = v

    # ---- bookkeeping (replay.py:115-142) ----
    num_added = property(lambda self: self._count, doc='items added since construction / reset')
    size = property(lambda self: min(self._count, self._cap), doc='items currently held')
    capacity = property(lambda self: self._cap, doc='ring size')

    def reset(self) -> None:
        self._count = 0

    def get_state(self) -> Mapping[Text, Any]:
        ring = None if self._ring is None else {k: v.cpu() for k, v in self._ring.items()}
        return {'num_added': self._count, 'storage': ring, 'priorities': self._prio}

    def set_state(self, state: Mapping[Text, Any]) -> None:
        self._count = state['num_added']
        self._ring = None if state['storage'] is None else {k: v.to(self._dev) for k, v in state['storage'].items()}
        self._prio = state['priorities']
