"""ctypes binding of libmzlearner_hip.so (C ABI: include/mzlearner.h) and `HipLearner`, the learner step of the MLP nets and -- round 5 --
of the board-game conv nets (`MuZeroBoardGameNet`: residual towers with train-mode BatchNorm) on hand-written gfx950 kernels (SURVEY 8 f2:
`calc_loss` pipeline.py:541-612 + backward + clip + Adam / MultiStepLR :238-255).

`HipLearner` owns ONE flat float32 device tensor holding master weights, gradient, exp_avg and exp_avg_sq; the network's
parameters are views into it (so `network.state_dict()` / checkpoints keep the reference's layout, pipeline.py:224-230), and
the kernels read sampled items straight out of the HBM replay ring by row index.  There is no PyTorch autograd and no ATen
kernel on the step; torch is the allocator, the stream and -- for a data-parallel learner -- the RCCL all-reduce of the flat
gradient between `grad()` and `apply()`.  No CPU fallback: without the library or a GPU the constructor raises."""
import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libmzlearner_hip.so')

# every symbol include/mzlearner.h declares (tests/test_abi.py checks the library exports all of them)
ABI_SYMBOLS = ['mzl_last_error', 'mzl_create', 'mzl_destroy', 'mzl_num_params', 'mzl_grad_floats', 'mzl_num_tensors', 'mzl_tensor_info',
               'mzl_num_buffers', 'mzl_num_running', 'mzl_buffer_info', 'mzl_bind_buffers', 'mzl_bind', 'mzl_commit', 'mzl_grad', 'mzl_apply',
               'mzl_replay_scratch_doubles', 'mzl_replay_sample', 'mzl_replay_update_priorities', 'mzl_replay_set_error_counters']
NET_MLP, NET_BOARD, NET_ATARI = 0, 1, 2


class LearnerError(RuntimeError):
    pass


class MzlConfig(C.Structure):
    _fields_ = [('in_dim', C.c_int32), ('num_actions', C.c_int32), ('num_planes', C.c_int32), ('hidden_dim', C.c_int32),
                ('value_support_size', C.c_int32), ('reward_support_size', C.c_int32), ('unroll_steps', C.c_int32), ('max_batch', C.c_int32),
                ('grad_slices', C.c_int32), ('net_kind', C.c_int32), ('in_channels', C.c_int32), ('board_h', C.c_int32), ('board_w', C.c_int32),
                ('num_res_blocks', C.c_int32)]


class MzlReplayDraw(C.Structure):
    _fields_ = [('d_priority', C.c_void_p), ('d_num_added', C.c_void_p), ('capacity', C.c_int64), ('priority_exponent', C.c_double),
                ('importance_sampling_exponent', C.c_double), ('seed', C.c_uint64), ('draw', C.c_uint64), ('batch', C.c_int32), ('d_index', C.c_void_p),
                ('d_weights', C.c_void_p), ('d_scratch', C.c_void_p)]


class MzlBatch(C.Structure):
    _fields_ = [('d_state', C.c_void_p), ('d_action', C.c_void_p), ('d_pi_prob', C.c_void_p), ('d_value', C.c_void_p), ('d_reward', C.c_void_p),
                ('d_index', C.c_void_p), ('d_weights', C.c_void_p), ('d_loss', C.c_void_p), ('d_priorities', C.c_void_p), ('batch', C.c_int32),
                ('state_is_int8', C.c_int32), ('action_bytes', C.c_int32)]


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LearnerError(f'{LIB_PATH} not found: build it with `python -m muzero_amd.build` (hipcc, gfx950). The HIP learner has no CPU fallback.')
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.mzl_last_error.restype = C.c_char_p
    L.mzl_create.argtypes = [C.POINTER(MzlConfig), C.c_int, C.POINTER(vp)]
    L.mzl_destroy.argtypes = [vp]
    L.mzl_num_params.argtypes = [vp]
    L.mzl_num_params.restype = i64
    L.mzl_grad_floats.argtypes = [vp]
    L.mzl_grad_floats.restype = i64
    L.mzl_tensor_info.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)]
    L.mzl_num_tensors.argtypes = [vp]
    L.mzl_num_tensors.restype = i32
    L.mzl_num_buffers.argtypes = [vp]
    L.mzl_num_buffers.restype = i32
    L.mzl_num_running.argtypes = [vp]
    L.mzl_num_running.restype = i64
    L.mzl_buffer_info.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(i64), C.POINTER(i32)]
    L.mzl_bind_buffers.argtypes = [vp, vp, vp]
    L.mzl_bind.argtypes = [vp, vp, vp, vp, vp]
    L.mzl_commit.argtypes = [vp, vp]
    L.mzl_grad.argtypes = [vp, C.POINTER(MzlBatch), vp]
    L.mzl_apply.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, i64, vp]
    L.mzl_replay_scratch_doubles.argtypes = [i64]
    L.mzl_replay_scratch_doubles.restype = i64
    L.mzl_replay_sample.argtypes = [C.POINTER(MzlReplayDraw), vp]
    L.mzl_replay_update_priorities.argtypes = [vp, i64, vp, vp, i32, vp, vp]
    L.mzl_replay_set_error_counters.argtypes = [vp]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise LearnerError(f'libmzlearner_hip: {load_library().mzl_last_error().decode()} (status {rc})')


class _OptimizerView:
    """What `run_training` (pipeline.py:170-286) holds as `optimizer`: the checkpoint's 'optimizer' entry in torch.optim.Adam's format."""

    def __init__(self, hl):
        self.hip_learner = hl

    def state_dict(self):
        return self.hip_learner.optimizer_state_dict()

    def load_state_dict(self, sd):
        self.hip_learner.load_optimizer_state_dict(sd)

    @property
    def param_groups(self):
        return self.hip_learner.optimizer_state_dict()['param_groups']


class _SchedulerView:
    """... and as `lr_scheduler` (MultiStepLR): state for the checkpoint, `get_last_lr` for the metrics."""

    def __init__(self, hl):
        self.hip_learner = hl

    def state_dict(self):
        return self.hip_learner.lr_scheduler_state_dict()

    def load_state_dict(self, sd):
        self.hip_learner.steps = int(sd.get('last_epoch', self.hip_learner.steps))

    def get_last_lr(self):
        return [self.hip_learner.current_lr()]

    def step(self):  # (the schedule advances inside HipLearner.apply)
        pass


def conv_learner_flops(input_shape, num_actions: int, num_res_blocks: int, num_planes: int, unroll_steps: int) -> float:
    """Algorithmic FLOPs (2 x multiply-accumulate) of ONE sample's update of a MuZeroBoardGameNet: forward + weight gradient of every 3 x 3
    conv of the K-step unroll (pipeline.py:575-592; the dynamics net's first conv over its num_planes + num_actions inputs, network.py:440-446)
    + data gradient wherever an input needs one (not the observation, not the constant action planes).  Heads and elementwise work are not
    counted (< 0.1 %)."""
    c0, h, w = input_shape
    P, A, R, K, hw = num_planes, num_actions, num_res_blocks, unroll_steps, h * w
    tower = 2 * R * P * P * 9 * hw
    fwd = (c0 * P * 9 * hw + tower) + K * ((P + A) * P * 9 * hw + tower) + K * tower
    dgrad = tower + K * (P * P * 9 * hw + tower) + K * tower
    return 2.0 * (2 * fwd + dgrad)


def atari_learner_flops(input_shape, num_actions: int, num_res_blocks: int, num_planes: int, unroll_steps: int) -> float:
    """conv_learner_flops for a MuZeroAtariNet (network.py:312-353 of the reference: conv_1 (stride 2) -> two 128-plane blocks at 48 x 48 ->
    conv_2 (stride 2) -> two blocks at 24 x 24 -> pool -> two blocks at 12 x 12 -> pool; dynamics / prediction towers on 6 x 6).  Algorithmic:
    the 14 x 14 halo tiles the kernels actually convolve (1.36 x the positions of the two large stages) are not counted."""
    c0, h, w = input_shape
    P, A, R, K = num_planes, num_actions, num_res_blocks, unroll_steps
    s1, s2, s3, s4 = (h // 2) * (w // 2), (h // 4) * (w // 4), (h // 8) * (w // 8), (h // 16) * (w // 16)
    rep_first = c0 * 128 * 9 * s1  # forward + weight gradient only
    rep_rest = 4 * 128 * 128 * 9 * s1 + 128 * P * 9 * s2 + 4 * P * P * 9 * s2 + 4 * P * P * 9 * s3
    tower = 2 * R * P * P * 9 * s4
    fwd = rep_first + rep_rest + K * ((P + A) * P * 9 * s4 + tower) + K * tower
    dgrad = rep_rest + K * (P * P * 9 * s4 + tower) + K * tower
    return 2.0 * (2 * fwd + dgrad)


class HipLearner:
    """The learner step of `run_training` (pipeline.py:238-255) for a `MuZeroMLPNet`, a `MuZeroBoardGameNet` or a `MuZeroAtariNet` on the GPU, as HIP kernels.
    (Conv nets: the BatchNorm layers run in TRAIN mode whatever `network.training` says -- a learner step is a training step -- and their
    running statistics, `num_batches_tracked` included, are module buffers re-pointed at the learner's flat buffer vectors.)

    `network`'s parameters are re-pointed at views of the learner's flat weight vector: after every `apply()` the module holds the
    new weights (its `state_dict()` is the checkpoint's 'network' entry as before).  Adam / MultiStepLR follow torch's definitions
    (`torch.optim.Adam(lr, weight_decay)` with L2-in-gradient, `MultiStepLR(milestones, gamma)`): `optimizer_state_dict()` returns
    what `torch.optim.Adam.state_dict()` would."""

    def __init__(self, network, device, unroll_steps: int, max_batch: int, lr: float, weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8,
                 milestones: Sequence[int] = (), gamma: float = 0.1, clip_grad: bool = False, max_grad_norm: float = 40.0, grad_slices: Optional[int] = None):
        self._h = C.c_void_p()
        if not torch.cuda.is_available():
            raise LearnerError('HipLearner needs a GPU (no CPU fallback)')
        L = load_library()
        self.device = torch.device(device)
        if self.device.type == 'cuda' and self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device() if torch.cuda.is_available() else 0)
        spec = network.planner_spec()
        if spec['kind'] not in ('mlp', 'board', 'atari'):
            raise LearnerError(f"HipLearner covers MuZeroMLPNet, MuZeroBoardGameNet and MuZeroAtariNet, not {spec['kind']!r}")
        self.kind = spec['kind']
        tiles = (max_batch + 15) // 16
        if self.kind in ('board', 'atari'):
            grad_slices = 1
        if grad_slices is None:
            # long reductions (large batches): the weight-gradient kernel runs one 8-wave workgroup per (layer, slice) -- 8 unrolled layers
            # with grad_slices slices each, 2 representation layers with grad_slices / K -- and all of them should be resident at once
            cus = torch.cuda.get_device_properties(device).multi_processor_count if torch.cuda.is_available() else 256
            grad_slices = 1 if tiles * unroll_steps < 256 else max(2, min(64, int(cus / (8.0 + 2.0 / unroll_steps)), tiles * unroll_steps // 8))
        in_dim = int(np.prod(spec['input_shape']))
        if self.kind in ('board', 'atari'):
            c0, bh, bw = spec['input_shape']
            cfg = MzlConfig(in_dim, spec['num_actions'], spec['num_planes'], 1, spec['value_support_size'], spec['reward_support_size'], unroll_steps, max_batch, 1,
                            NET_BOARD if self.kind == 'board' else NET_ATARI, c0, bh, bw, spec['num_res_blocks'])
        else:
            cfg = MzlConfig(in_dim, spec['num_actions'], spec['num_planes'], spec['hidden_dim'], spec['value_support_size'], spec['reward_support_size'],
                            unroll_steps, max_batch, grad_slices, NET_MLP, 0, 0, 0, 0)
        _check(L.mzl_create(C.byref(cfg), self.device.index or 0, C.byref(self._h)))
        self.K, self.A, self.in_dim, self.max_batch = unroll_steps, spec['num_actions'], in_dim, max_batch
        self.total = int(L.mzl_num_params(self._h))
        gfl = int(L.mzl_grad_floats(self._h))
        self.flat = torch.zeros(3 * self.total + gfl, dtype=torch.float32, device=self.device)
        self.params, self.exp_avg, self.exp_avg_sq = (self.flat[i * self.total:(i + 1) * self.total] for i in range(3))
        self.grads = self.flat[3 * self.total:]
        self.grad_flat = self.grads[:self.total]  # the complete gradient after grad() (slice 0)
        self.views, self.grad_views = {}, {}
        module_shapes = {k: tuple(p.shape) for k, p in network.named_parameters()}
        for i in range(int(L.mzl_num_tensors(self._h))):
            name, off, rows, cols = C.c_char_p(), C.c_int64(), C.c_int32(), C.c_int32()
            _check(L.mzl_tensor_info(self._h, i, C.byref(name), C.byref(off), C.byref(rows), C.byref(cols)))
            key = name.value.decode()
            n = rows.value * (cols.value if cols.value else 1)
            shape = module_shapes.get(key, (rows.value, cols.value) if cols.value else (rows.value,))  # (conv weights: [cout, cin, 3, 3] / [oc, P, 1, 1])
            if int(np.prod(shape)) != n:
                raise LearnerError(f'{key}: the module holds {shape}, the learner {n} elements')
            self.views[key] = self.params[off.value:off.value + n].view(shape)
            self.grad_views[key] = self.grad_flat[off.value:off.value + n].view(shape)
        _check(L.mzl_bind(self._h, self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()))
        # BatchNorm buffers (conv nets): flat running statistics + one num_batches_tracked per layer, written by the kernels
        self.buffer_views = {}
        nbuf = int(L.mzl_num_buffers(self._h))
        self.running = torch.zeros(max(1, int(L.mzl_num_running(self._h))), dtype=torch.float32, device=self.device)
        self.num_batches = torch.zeros(max(1, nbuf), dtype=torch.int64, device=self.device)
        for i in range(nbuf):
            name, off, cnt = C.c_char_p(), C.c_int64(), C.c_int32()
            _check(L.mzl_buffer_info(self._h, i, C.byref(name), C.byref(off), C.byref(cnt)))
            key, o, c = name.value.decode(), off.value, cnt.value
            self.buffer_views[key + '.running_mean'] = self.running[o:o + c]
            self.buffer_views[key + '.running_var'] = self.running[o + c:o + 2 * c]
            self.buffer_views[key + '.num_batches_tracked'] = self.num_batches[i]
        if nbuf:
            _check(L.mzl_bind_buffers(self._h, self.running.data_ptr(), self.num_batches.data_ptr()))
        self.lr_init, self.weight_decay, self.betas, self.eps = float(lr), float(weight_decay), tuple(betas), float(eps)
        self.milestones, self.gamma = sorted(int(m) for m in milestones), float(gamma)
        self.clip_grad, self.max_grad_norm = bool(clip_grad), float(max_grad_norm)
        self.steps = 0
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.priorities = torch.zeros(max_batch, dtype=torch.float32, device=self.device)
        self._ones = torch.ones(max_batch, dtype=torch.float32, device=self.device)
        self._iota = torch.arange(max_batch, dtype=torch.int64, device=self.device)
        self.network = network
        self.optimizer, self.lr_scheduler = _OptimizerView(self), _SchedulerView(self)
        self.adopt(network)

    # ---- weights ----
    def adopt(self, network) -> None:
        """Copy `network`'s weights into the flat vector and make its parameters views of it."""
        sd = network.state_dict()
        mine = set(self.views) | set(self.buffer_views)
        if set(sd) != mine:
            raise LearnerError(f'state_dict keys do not match the learner\'s network: {sorted(set(sd) ^ mine)}')
        with torch.no_grad():
            for k, v in sd.items():
                dst = self.views[k] if k in self.views else self.buffer_views[k]
                if tuple(v.shape) != tuple(dst.shape):
                    raise LearnerError(f'{k}: shape {tuple(v.shape)} != {tuple(dst.shape)}')
                dst.copy_(v.to(self.device, dst.dtype))
            for k, p in network.named_parameters():
                p.data = self.views[k]
            for k, b in network.named_buffers():
                if k in self.buffer_views:
                    b.data = self.buffer_views[k]
        self.network = network
        self.commit()

    def _param_versions(self):
        return tuple(p._version for p in self.network.parameters())

    def load_state_dict(self, sd) -> None:
        with torch.no_grad():
            for k, v in sd.items():
                dst = self.views[k] if k in self.views else self.buffer_views[k]
                dst.copy_(torch.as_tensor(v).to(self.device, dst.dtype))
        self.commit()

    def state_dict(self):
        out = {k: v.detach().clone() for k, v in self.views.items()}
        out.update({k: v.detach().clone() for k, v in self.buffer_views.items()})
        return out

    def _bump_epoch(self) -> None:
        # torch's version counters do not see writes through the flat vector's views or by the kernels: the module's inference engine
        # (network.MuZeroNet._weights_version) rebinds on this epoch (load_state_dict / adopt left it on the old weights)
        self.network._mz_weights_epoch = getattr(self.network, '_mz_weights_epoch', 0) + 1

    def commit(self) -> None:
        """Rebuild the MFMA operand copies from the master weights (after anything other than `apply` wrote them)."""
        _check(load_library().mzl_commit(self._h, self._stream()))
        self._seen_versions = self._param_versions()
        self._bump_epoch()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- optimizer bookkeeping in torch's terms ----
    def current_lr(self, step: Optional[int] = None) -> float:
        """MultiStepLR: the rate of update number `step` (0-based count of updates done before it)."""
        s = self.steps if step is None else step
        return self.lr_init * self.gamma ** sum(1 for m in self.milestones if m <= s)

    def _param_names(self):
        return [k for k, _ in self.network.named_parameters()]  # torch.optim's parameter indices follow this order

    def optimizer_state_dict(self):
        names = self._param_names()
        state = {}
        for i, k in enumerate(names):
            v = self.views[k]
            off = v.data_ptr() - self.params.data_ptr()
            off //= 4
            state[i] = dict(step=torch.tensor(float(self.steps)), exp_avg=self.exp_avg[off:off + v.numel()].view(v.shape).clone(),
                            exp_avg_sq=self.exp_avg_sq[off:off + v.numel()].view(v.shape).clone())
        group = dict(lr=self.current_lr(), betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, amsgrad=False, maximize=False,
                     params=list(range(len(names))))
        return dict(state=state, param_groups=[group])

    def load_optimizer_state_dict(self, osd) -> None:
        names = self._param_names()
        for i, k in enumerate(names):
            st = osd['state'].get(i)
            if st is None:
                continue
            v = self.views[k]
            off = (v.data_ptr() - self.params.data_ptr()) // 4
            self.exp_avg[off:off + v.numel()].copy_(st['exp_avg'].reshape(-1).to(self.device))
            self.exp_avg_sq[off:off + v.numel()].copy_(st['exp_avg_sq'].reshape(-1).to(self.device))
            self.steps = int(float(st['step']))

    def lr_scheduler_state_dict(self):
        return dict(milestones={m: 1 for m in self.milestones}, gamma=self.gamma, last_epoch=self.steps, _step_count=self.steps + 1,
                    base_lrs=[self.lr_init], _last_lr=[self.current_lr()])

    # ---- the step ----
    def grad(self, ring, index: Optional[torch.Tensor], weights: Optional[torch.Tensor], batch: int):
        """Loss and gradient of one batch (pipeline.py:241-244).  `ring`: mapping field -> device tensor [rows, ...] (the replay's
        storages, or a stacked batch with index=None); `index`: int64 device tensor [batch] of rows; `weights`: float32 [batch] or None
        (uniform replay: ones).  Fills `self.grad_flat`, `self.loss`, `self.priorities[:batch]`; enqueues only."""
        if batch < 1 or batch > self.max_batch:
            raise LearnerError(f'batch {batch} outside [1, max_batch = {self.max_batch}]')
        if self._param_versions() != self._seen_versions:
            # torch wrote the parameters (network.load_state_dict on a checkpoint -- the reference's resume path, pipeline.py:810-817 -- or an
            # in-place edit): the master weights are those tensors, the kernels' operand copies are rebuilt from them
            self.commit()
        st, ac = ring['state'], ring['action']
        if st.dtype not in (torch.float32, torch.int8) or not st.is_contiguous():
            raise LearnerError(f'state storage must be contiguous float32 or int8, got {st.dtype}')
        if ac.dtype not in (torch.int8, torch.int16) or not ac.is_contiguous():
            raise LearnerError(f'action storage must be contiguous int8 or int16, got {ac.dtype}')
        if int(np.prod(st.shape[1:])) != self.in_dim or tuple(ac.shape[1:]) != (self.K,) or tuple(ring['pi_prob'].shape[1:]) != (self.K, self.A):
            raise LearnerError('replay item shapes do not match the learner')
        for f in ('pi_prob', 'value', 'reward'):
            if ring[f].dtype != torch.float32 or not ring[f].is_contiguous():
                raise LearnerError(f'{f} storage must be contiguous float32')
        if index is None:
            index = self._iota[:batch]
        # every pointer handed to the kernels must be memory of the learner's GPU: a host-resident replay (PrioritizedReplay's default
        # device) would be read as a device address
        for name, t in (('state', st), ('action', ac), ('pi_prob', ring['pi_prob']), ('value', ring['value']), ('reward', ring['reward']),
                        ('index', index), ('weights', self._ones if weights is None else weights)):
            if t.device != self.device:
                raise LearnerError(f'{name} lives on {t.device}, the learner on {self.device}: build the replay with device=\'cuda\' '
                                   f'(muzero_amd.replay.PrioritizedReplay(..., device=\'cuda\')) and pass device tensors')
        if index.dtype != torch.int64 or index.numel() != batch:
            raise LearnerError('index must be an int64 tensor of `batch` rows')
        w = self._ones if weights is None else weights
        if w.dtype != torch.float32 or w.numel() < batch:
            raise LearnerError('weights must be a float32 tensor of `batch` entries')
        b = MzlBatch(st.data_ptr(), ac.data_ptr(), ring['pi_prob'].data_ptr(), ring['value'].data_ptr(), ring['reward'].data_ptr(),
                     index.data_ptr(), w.data_ptr(), self.loss.data_ptr(), self.priorities.data_ptr(), batch,
                     1 if st.dtype == torch.int8 else 0, ac.element_size())
        _check(load_library().mzl_grad(self._h, C.byref(b), self._stream()))
        self._keep = (ring, index, w)  # alive until the next call (the kernels read them asynchronously)
        if self.buffer_views:
            self._bump_epoch()  # (train-mode BatchNorm: the forward pass has updated the running statistics the planner folds into its weights)
        return self.loss, self.priorities[:batch]

    def apply(self, clip: Optional[bool] = None) -> None:
        """clip_grad_norm_ (if configured) + optimizer.step() + lr_scheduler.step() (pipeline.py:246-250)."""
        clip = self.clip_grad if clip is None else clip
        lr = self.current_lr()
        self.steps += 1
        _check(load_library().mzl_apply(self._h, lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.max_grad_norm if clip else 0.0,
                                        self.steps, self._stream()))
        self._bump_epoch()

    def step(self, ring, index, weights, batch: int, allreduce: bool = True):
        """One update.  With an initialised multi-rank process group the flat gradient is averaged over the ranks in ONE all-reduce
        (RCCL; xGMI rings are per-link bound, so one 1-30 MB transfer beats per-tensor collectives) between gradient and update."""
        loss, prio = self.grad(ring, index, weights, batch)
        if allreduce:
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.all_reduce(self.grad_flat, op=dist.ReduceOp.SUM)
                self.grad_flat.div_(dist.get_world_size())
        self.apply()
        return loss, prio

    def step_transitions(self, transitions, weights=None):
        """One update on a stacked batch (`Transition` of arrays / tensors, as `replay.sample` returns them)."""
        def dev(x, dt):
            t = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x))
            return t.to(self.device, dt).contiguous()

        st = transitions.state
        st_dt = torch.int8 if (torch.is_tensor(st) and st.dtype == torch.int8) or (not torch.is_tensor(st) and np.asarray(st).dtype == np.int8) else torch.float32
        B = int(st.shape[0])
        ring = dict(state=dev(st, st_dt).reshape(B, -1), action=dev(transitions.action, torch.int16 if self.A > 128 else torch.int8),
                    pi_prob=dev(transitions.pi_prob, torch.float32), value=dev(transitions.value, torch.float32), reward=dev(transitions.reward, torch.float32))
        w = None if weights is None else dev(weights, torch.float32)
        return self.step(ring, None, w, B)

    def close(self) -> None:
        if self._h:
            load_library().mzl_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
