"""Replay for the learner side of the path -- the reference's `replay.py:27-166` API
(`Transition`, `PrioritizedReplay.add / get / sample / update_priorities / size / num_added / capacity / reset /
get_state / set_state`) over a structure-of-arrays ring buffer instead of a Python list of snappy-compressed tuples.

Storage is one preallocated tensor per `Transition` field (`state [cap, *shape]`, `action int8 [cap, K]`,
`pi_prob f32 [cap, K, A]`, `value f32 [cap, K]`, `reward f32 [cap, K]`) on a torch device: with `device='cuda'` the
ring lives in HBM, `sample_tensors` hands the learner device tensors without a host round trip, and the planner's
device epilogue (`Planner.attach_replay`) writes finished transitions straight into it.  States are stored in the dtype
of the first item unless `state_dtype` says otherwise: float32 Atari-sized states (8x96x96) cost 295 KB each, so 10^6 of
them do NOT fit even 288 GB -- pass `state_dtype=torch.float16` / `torch.uint8` for image states (converted back to
float32 in `sample_tensors`); the constructor-time size check refuses a ring that exceeds the free memory of its device
instead of letting the process die.  With the default `device='cpu'` it is a drop-in for the reference class.

Thread safety: the collector thread adds while the learner thread samples (pipeline.py:491-538 vs :238-255).  The
reference swaps one tuple per slot atomically; here the five field writes of an add and the gather of a sample hold one
lock, so a batch never contains a torn transition written by a HOST writer.  While a planner's device epilogue is attached
the DEVICE owns the write cursor and the priorities: the host never writes the counter (add / add_batch / reset raise),
`update_priorities` scatters only the touched entries on the device, and what the host reads as `num_added` is the
COMMITTED count -- published by the epilogue kernel after every slot below it has been filled (mz_env.h, k_epilogue) -- so
a batch never contains a slot that is still being written for the first time.  (Once the ring has wrapped, a slot may be
overwritten while it is gathered -- the reference's actors / learner threads have the same window per tuple swap.)
Sampling semantics are the reference's,
draw for draw: uniform replay draws `random_state.uniform(0, size, batch).astype(int64)` (`replay.py:87-89`),
prioritized replay draws from the process-global `np.random.choice` (`replay.py:90-98`; an upstream quirk kept on
purpose) with importance weights `((1/size) / p_i)^beta / max`."""
import threading
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
                 device='cpu', state_dtype: Optional[torch.dtype] = None):
        if capacity <= 0:
            raise ValueError(f'Expect capacity to be a positive integer, got {capacity}')
        self.structure = TransitionStructure
        self._cap, self._count = int(capacity), 0
        self._rng = random_state
        self._alpha, self._beta = priority_exponent, importance_sampling_exponent
        self._prio = np.zeros(self._cap, dtype=np.float32)
        self._dev = torch.device(device)
        self._state_dtype = state_dtype
        self._ring = None  # field -> tensor [capacity, ...]; allocated on the first add, when the shapes are known
        self._lock = threading.Lock()
        self._attached = None  # device-side (priority, num_added) tensors while a planner epilogue writes into the ring

    # ---- storage ----
    @staticmethod
    def _check_priorities(pr) -> None:
        pr = np.asarray(pr, np.float64)
        if not np.isfinite(pr).all() or (pr < 0.0).any():
            raise ValueError('Priority must be finite and positive.')

    def allocate(self, shapes: Mapping[Text, Tuple[int, ...]], dtypes: Optional[Mapping[Text, torch.dtype]] = None) -> None:
        """Allocate the ring for per-item field shapes (done implicitly by the first add; explicitly before a planner
        epilogue is attached).  Raises MemoryError if `capacity` items do not fit the free memory of the device."""
        defaults = dict(state=torch.float32, action=torch.int8, pi_prob=torch.float32, value=torch.float32, reward=torch.float32)
        defaults.update(dtypes or {})
        if self._state_dtype is not None:
            defaults['state'] = self._state_dtype
        need = sum(int(np.prod(shapes[f], dtype=np.int64)) * torch.empty(0, dtype=defaults[f]).element_size() for f in self.FIELDS) * self._cap
        if self._dev.type == 'cuda':
            free = torch.cuda.mem_get_info(self._dev)[0]
        else:
            # host rings are committed lazily (torch.zeros maps zero pages), exactly like the reference's list that grows
            # as items arrive: refuse only what can never fit, i.e. more than the machine's physical memory
            import os

            try:
                free = os.sysconf('SC_PHYS_PAGES') * os.sysconf('SC_PAGE_SIZE') / 0.9
            except (ValueError, OSError, AttributeError):
                free = float('inf')
        if need > 0.9 * free:
            raise MemoryError(f'replay of {self._cap} items needs {need / 2**30:.1f} GiB on {self._dev} but only {free / 2**30:.1f} GiB are free: '
                              f'lower the capacity or store states in a narrower dtype (state_dtype=torch.float16 / torch.uint8)')
        self._ring = {f: torch.zeros((self._cap,) + tuple(shapes[f]), dtype=defaults[f], device=self._dev) for f in self.FIELDS}

    def _allocate(self, item: Transition) -> None:
        arrs = [np.asarray(x) for x in item]
        self.allocate({f: a.shape for f, a in zip(self.FIELDS, arrs)},
                      {f: torch.from_numpy(np.zeros(1, a.dtype)).dtype for f, a in zip(self.FIELDS, arrs)})

    def _check_shapes(self, item_shapes) -> None:
        for f, shp in zip(self.FIELDS, item_shapes):
            if tuple(shp) != tuple(self._ring[f].shape[1:]):
                raise ValueError(f'Transition.{f} has shape {tuple(shp)}, the replay holds {tuple(self._ring[f].shape[1:])}')

    def _host_writer_only(self, what: str) -> None:
        if self._attached is not None:
            raise RuntimeError(f'PrioritizedReplay.{what}: a planner device epilogue is attached and owns the write cursor of this ring; '
                               f'detach it first (Planner.detach_replay + PrioritizedReplay.detach_device_writer)')

    def add(self, item: Transition, priority: float) -> None:
        """One item into the ring slot `num_added % capacity` (replay.py:67-75)."""
        self._check_priorities(priority)
        with self._lock:
            self._host_writer_only('add')
            if self._ring is None:
                self._allocate(item)
            arrs = [np.ascontiguousarray(x) for x in item]
            self._check_shapes([a.shape for a in arrs])
            slot = self._count % self._cap
            for name, a in zip(self.FIELDS, arrs):
                self._ring[name][slot] = torch.from_numpy(a).to(self._ring[name].dtype)
            self._prio[slot] = priority
            self._count += 1

    def add_batch(self, items: Transition, priorities: Sequence[float]) -> None:
        """n items at once (fields stacked on axis 0): the form the device-resident actor produces."""
        self._check_priorities(priorities)
        n = len(priorities)
        with self._lock:
            self._host_writer_only('add_batch')
            if self._ring is None:
                self._allocate(Transition(*[np.asarray(x)[0] for x in items]))
            self._check_shapes([tuple(x.shape[1:]) for x in items])
            slots = (self._count + np.arange(n)) % self._cap
            tslots = torch.from_numpy(slots).to(self._dev)
            for name, x in zip(self.FIELDS, items):
                x = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x))
                self._ring[name][tslots] = x.to(self._dev, dtype=self._ring[name].dtype)
            self._prio[slots] = np.asarray(priorities, np.float32)
            self._count += n

    def get(self, indices: Sequence[int]) -> List[Transition]:
        """Items by index (replay.py:77-79)."""
        with self._lock:
            self._sync_attached()
            return [Transition(*[self._ring[f][int(i)].cpu().numpy() for f in self.FIELDS]) for i in indices]

    # ---- device epilogue (Planner.attach_replay): the planner writes items and priorities, this class only counts ----
    def attach_device_writer(self):
        """Tensors a device-side writer needs besides the ring: float32 priorities [capacity] and the int64 COMMITTED
        `num_added` counter, both on the replay's device, initialised from the host's bookkeeping.  From here on the device
        owns both: the host only reads the counter (the writer publishes it after the slots below it are filled) and
        scatters single priorities (`update_priorities`); it never writes the counter or the whole priority array, so
        nothing the GPU wrote in the meantime can be lost."""
        if self._ring is None:
            raise RuntimeError('allocate() the ring before attaching a device writer')
        if self._attached is not None:
            # SINGLE WRITER: each planner reserves slots from its own cursor (seeded from this counter) and publishes the counter by
            # overwriting it -- two planners on one ring would reserve the same slots and move the counter backwards
            raise RuntimeError('PrioritizedReplay: a device writer is already attached; one planner epilogue per replay '
                               '(give every actor GPU / planner its own replay shard, or detach the first writer)')
        prio = torch.from_numpy(self._prio.copy()).to(self._dev)
        count = torch.tensor([self._count], dtype=torch.int64, device=self._dev)
        self._attached = (prio, count)
        return self._attached

    def detach_device_writer(self) -> None:
        """Take the bookkeeping back to the host (after Planner.detach_replay, which drains the planner's stream)."""
        with self._lock:
            self._sync_attached()
            self._attached = None

    def _sync_attached(self) -> None:
        """read-only refresh of the host's view from the device-owned counter and priorities"""
        if self._attached is not None:
            prio, count = self._attached
            self._count = int(count.item())
            self._prio = prio.cpu().numpy()

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
        """Like `sample`, but the batch stays on the replay's device as torch tensors (no host copy).  States stored in a
        narrow dtype come back as float32."""
        with self._lock:
            if self._attached is not None:
                self._count = int(self._attached[1].item())
                if self._alpha != 0:
                    self._prio = self._attached[0].cpu().numpy()
            picks, is_w = self._draw(batch_size)
            tpicks = torch.from_numpy(picks).to(self._dev)
            batch = [self._ring[f].index_select(0, tpicks) for f in self.FIELDS]
        if batch[0].dtype != torch.float32 and self._state_dtype is not None:
            batch[0] = batch[0].to(torch.float32)
        return Transition(*batch), picks, is_w

    def sample_indices(self, batch_size: int):
        """The draw of `sample` without the gather: (indices int64 [B], importance weights float32 [B], the ring's storages).  For a
        learner that reads its batch out of the ring itself (hip_learner.HipLearner: the kernels gather by index)."""
        with self._lock:
            if self._attached is not None:
                self._count = int(self._attached[1].item())
                if self._alpha != 0:
                    self._prio = self._attached[0].cpu().numpy()
            picks, is_w = self._draw(batch_size)
        return picks, is_w, self._ring

    def device_sampler(self, seed: int = 0) -> 'DeviceSampler':
        """Sampling and priority updates ON THE DEVICE for a ring whose bookkeeping lives there (a device writer attached: the planner's
        epilogue): see `DeviceSampler`.  The host path above stays the draw-for-draw parity mode."""
        return DeviceSampler(self, seed)

    def sample(self, batch_size: int) -> Tuple[Transition, np.ndarray, np.ndarray]:
        """A batch with replacement (replay.py:81-104): numpy arrays stacked on axis 0, the indices, the IS weights."""
        batch, picks, is_w = self.sample_tensors(batch_size)
        return Transition(*[x.cpu().numpy() for x in batch]), picks, is_w

    def update_priorities(self, indices: Sequence[int], priorities: Sequence[float]) -> None:
        """replay.py:106-113 (a repeated index keeps the last value, like the reference's loop)."""
        pr = np.asarray(priorities)
        if not np.isfinite(pr).all() or (pr < 0.0).any():
            raise ValueError('Priorities must be finite and positive.')
        with self._lock:
            if self._attached is not None:
                # the device owns the array: scatter only the touched entries (last value per repeated index), on the device
                ia = np.asarray(indices, np.int64)
                if ia.size and (ia.min() < 0 or ia.max() >= self._cap):
                    raise IndexError(f'priority index outside [0, {self._cap})')
                last = {int(i): float(v) for i, v in zip(indices, pr)}
                if last:
                    idx = torch.tensor(list(last.keys()), dtype=torch.int64, device=self._dev)
                    val = torch.tensor(list(last.values()), dtype=torch.float32, device=self._dev)
                    self._attached[0].index_copy_(0, idx, val)
                return
            for i, v in zip(indices, pr):
                self._prio[i] = v

    # ---- bookkeeping (replay.py:115-142) ----
    def _live_count(self) -> int:
        if self._attached is not None:
            self._count = int(self._attached[1].item())
        return self._count

    num_added = property(lambda self: self._live_count(), doc='items added since construction / reset')
    size = property(lambda self: min(self._live_count(), self._cap), doc='items currently held')
    capacity = property(lambda self: self._cap, doc='ring size')

    def reset(self) -> None:
        with self._lock:
            self._host_writer_only('reset')
            self._count = 0

    def get_state(self) -> Mapping[Text, Any]:
        self._sync_attached()
        ring = None if self._ring is None else {k: v.cpu() for k, v in self._ring.items()}
        return {'num_added': self._count, 'storage': ring, 'priorities': self._prio}

    def set_state(self, state: Mapping[Text, Any]) -> None:
        self._host_writer_only('set_state')  # (while attached the next read of the device counter would overwrite the restored state)
        self._count = state['num_added']
        self._ring = None if state['storage'] is None else {k: v.to(self._dev) for k, v in state['storage'].items()}
        self._prio = state['priorities']


class DeviceSampler:
    """`PrioritizedReplay.sample` / `update_priorities` (replay.py:81-113) without the host: the kernels of libmzlearner_hip.so
    (include/mzlearner.h, mzl_replay_sample / mzl_replay_update_priorities) read the committed item count and the priorities where the
    planner's device epilogue publishes them and leave indices and importance weights in HBM for `hip_learner.HipLearner.grad`.  No `.item()`,
    no copy of the priority array, nothing synchronises: a learner loop built on it enqueues and returns.

    Uniform replay (priority_exponent == 0, all launchers' default): index = floor(u * size), weights None (= ones).  Proportional replay:
    inverse-CDF picks on float64 prefix sums of priority ^ alpha -- np.random.choice's own method, with Philox uniforms keyed by
    (seed, draw number, sample) instead of the process-global MT19937 -- and importance weights ((1 / size) / p) ^ beta / max.  Same
    distribution as the reference, not the same draws: the host path of `PrioritizedReplay` remains the draw-for-draw mode."""

    def __init__(self, replay: PrioritizedReplay, seed: int = 0):
        from muzero_amd import hip_learner as _hl  # (loads libmzlearner_hip.so; raises without it: no CPU fallback)

        if replay._attached is None or replay._dev.type != 'cuda':
            raise RuntimeError('DeviceSampler needs a replay in HBM whose bookkeeping the device owns: PrioritizedReplay(device=\'cuda\') with a '
                               'planner epilogue attached (Planner.attach_replay) or attach_device_writer() called')
        self._hl, self._lib = _hl, _hl.load_library()
        self.replay, self.seed, self.draws = replay, int(seed) & (2 ** 64 - 1), 0
        self._prio, self._count = replay._attached
        dev, cap = replay._dev, replay._cap
        self._scratch = None if replay._alpha == 0 else torch.zeros(int(self._lib.mzl_replay_scratch_doubles(cap)), dtype=torch.float64, device=dev)
        self._owner = None
        self._bufs = {}
        # where the kernels count what the reference would have raised on (include/mzlearner.h, mzl_replay_set_error_counters): [0] draws from an
        # empty replay, [1] invalid priorities (skipped).  Read by `check_errors()` wherever the caller synchronises anyway.
        self._errors = torch.zeros(2, dtype=torch.int32, device=dev)

    def _bind_errors(self):
        self._hl._check(self._lib.mzl_replay_set_error_counters(self._errors.data_ptr()))

    def check_errors(self) -> None:
        """Raises what `PrioritizedReplay` would have raised at the call (replay.py:83-84, 106-110) -- later, at a point where the caller reads
        something back from the device anyway (run_training: its metrics line).  One device -> host copy of 8 bytes."""
        empty, bad = (int(x) for x in self._errors.tolist())
        if empty or bad:
            self._errors.zero_()
        if bad:
            raise ValueError('Priorities must be finite and positive.')  # (the reference's message, replay.py:110)
        if empty:
            raise RuntimeError('sample() from an empty replay')

    def _stream(self):
        import ctypes as C

        return C.c_void_p(torch.cuda.current_stream(self.replay._dev).cuda_stream)

    def sample(self, batch_size: int):
        """(indices int64 [B] on the device, importance weights float32 [B] on the device -- None for uniform replay --, the ring's storages).
        The caller must know the replay holds at least one item (the kernels cannot raise): check `replay.size` once, before the loop."""
        import ctypes as C

        rp = self.replay
        if self._attached_changed():
            raise RuntimeError('the replay\'s device writer was detached: build a new DeviceSampler')
        if batch_size not in self._bufs:
            self._bufs[batch_size] = (torch.zeros(batch_size, dtype=torch.int64, device=rp._dev), torch.ones(batch_size, dtype=torch.float32, device=rp._dev))
        idx, w = self._bufs[batch_size]
        d = self._hl.MzlReplayDraw(self._prio.data_ptr(), self._count.data_ptr(), rp._cap, float(rp._alpha), float(rp._beta), self.seed, self.draws, batch_size,
                                   idx.data_ptr(), w.data_ptr(), 0 if self._scratch is None else self._scratch.data_ptr())
        self.draws += 1
        self._bind_errors()
        self._hl._check(self._lib.mzl_replay_sample(C.byref(d), self._stream()))
        return idx, (None if rp._alpha == 0 else w), rp._ring

    def update_priorities(self, indices: torch.Tensor, priorities: torch.Tensor) -> None:
        """replay.py:106-113 on device tensors (of a repeated index the last value wins).  Values are validated where they are -- the kernel
        skips a non-finite or negative priority and counts it; `check_errors()` raises the reference's ValueError at the caller's next
        synchronisation point (a host-side check here would be the read-back this class exists to avoid)."""
        rp = self.replay
        if self._owner is None:
            self._owner = torch.zeros(rp._cap, dtype=torch.int32, device=rp._dev)
        n = int(indices.numel())
        if indices.dtype != torch.int64 or priorities.dtype != torch.float32 or priorities.numel() < n or indices.device != self._prio.device or priorities.device != self._prio.device:
            raise ValueError('update_priorities: int64 indices and float32 priorities on the replay\'s device')
        self._bind_errors()
        self._hl._check(self._lib.mzl_replay_update_priorities(self._prio.data_ptr(), rp._cap, indices.data_ptr(), priorities.data_ptr(), n,
                                                               self._owner.data_ptr(), self._stream()))

    def _attached_changed(self) -> bool:
        return self.replay._attached is None or self.replay._attached[0] is not self._prio
