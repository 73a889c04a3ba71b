"""Learner step of the pipeline: `calc_loss` (pipeline.py:541-629), `loss_func` (:615-629), the target
projection (util.py:48-59,96-116), `run_training` (:170-286) and `run_data_collector` (:491-538).

PyTorch-ROCm autograd over the `muzero_amd.network` modules (the same modules whose weights the HIP planner consumes);
what is MI355X-specific is the data path around it: batches come from the HBM-resident replay (`replay.sample_tensors`,
no host round trip) and, with one learner process per GPU, gradients are averaged with ONE flat all-reduce per step over
RCCL (`allreduce_gradients`: xGMI is point-to-point, so a single large ring transfer beats per-tensor collectives; the
biggest reference network is 30 MB of fp32 gradients).  Self-play itself never joins a collective.

Reference behaviours kept on purpose: the 0.5 gradient scale on the unrolled hidden state (pipeline.py:584), the 1/K
scale applied to the *gradient* of the batch-mean loss, not to the reported loss (:600), per-sample importance weights
(:597), priorities from the step-0 value error (:609), the reward loss also for board games (:589-590)."""
import queue
import time
from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from muzero_amd.network import MuZeroNet, logits_to_transformed_expected_value, signed_hyperbolic
from muzero_amd.pipeline import create_checkpoint
from muzero_amd.replay import PrioritizedReplay, Transition


# ---------------------------------------------------------------------------------------------------------------
# targets
# ---------------------------------------------------------------------------------------------------------------
def transform_to_2hot(scalar: torch.Tensor, min_value: float, max_value: float, num_bins: int) -> torch.Tensor:
    """Scalar -> categorical over `num_bins` equally spaced support points in [min_value, max_value]: the two bins that
    bracket the (clamped) scalar share the mass (semantics of util.py:48-59, including its 1e-5 in the interpolation
    denominator, which the recorded reference projections pin).  Written as two scatter-adds into a zero tensor."""
    span = max_value - min_value
    z = scalar.clamp(min_value, max_value)
    pos = (z - min_value) / span * (num_bins - 1)
    below, above = pos.floor(), pos.ceil()

    def support(i):
        return i / (num_bins - 1.0) * span + min_value

    w_below = (support(above) - z) / (support(above) - support(below) + 1e-5)
    out = torch.zeros(*z.shape, num_bins, dtype=z.dtype, device=z.device)
    out.scatter_add_(-1, below.long().unsqueeze(-1), w_below.unsqueeze(-1))
    out.scatter_add_(-1, above.long().unsqueeze(-1), (1 - w_below).unsqueeze(-1))
    return out


def scalar_to_categorical_probabilities(x: torch.Tensor, support_size: int) -> torch.Tensor:
    """util.py:96-116: signed_hyperbolic, then projection onto the integer support [-(S-1)/2, (S-1)/2]."""
    half = (support_size - 1) // 2
    return transform_to_2hot(signed_hyperbolic(x), -half, half, support_size)


def loss_func(prediction: torch.Tensor, target: torch.Tensor, mse: bool = False) -> torch.Tensor:
    """Per-sample loss of one head (pipeline.py:615-629): squared error for scalar heads, cross entropy against a SOFT
    target distribution for categorical heads and the policy.  Works on [B, ...] and on stacked [B, K, ...] inputs."""
    if prediction.shape != target.shape:
        raise ValueError(f'prediction {tuple(prediction.shape)} and target {tuple(target.shape)} differ in shape')
    if mse:
        return (prediction - target) ** 2
    return -(target * F.log_softmax(prediction, dim=-1)).sum(dim=-1)


class _ScaleGradient(torch.autograd.Function):
    """Identity in the forward pass; multiplies the incoming gradient by a constant in the backward pass."""

    @staticmethod
    def forward(ctx, x, factor):
        ctx.factor = factor
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad):
        return grad * ctx.factor, None


def scale_gradient(x: torch.Tensor, factor: float) -> torch.Tensor:
    return _ScaleGradient.apply(x, factor)


def _as_tensor(x, device, dtype):
    t = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x))
    return t.to(device=device, dtype=dtype, non_blocking=True)


def unroll(network: MuZeroNet, state: torch.Tensor, action: torch.Tensor):
    """K-step unroll of the learned model from the root observation (pipeline.py:575-592): returns the stacked head
    outputs policy logits [B, K, A], value [B, K, Sv], reward [B, K, Sr].  The gradient flowing back into each unrolled
    hidden state is halved (pipeline.py:584)."""
    pis, values, rewards = [], [], []
    hidden = network.represent(state)
    for k in range(action.shape[1]):
        pi_logits, value = network.prediction(hidden)
        hidden, reward = network.dynamics(hidden, action[:, k:k + 1])
        hidden = scale_gradient(hidden, 0.5)
        pis.append(pi_logits)
        values.append(value)
        rewards.append(reward)
    return torch.stack(pis, dim=1), torch.stack(values, dim=1), torch.stack(rewards, dim=1)


def loss_tensors(network: MuZeroNet, state, action, value_scalar, reward_scalar, pi_target, weights) -> Tuple[torch.Tensor, torch.Tensor]:
    """`calc_loss` on tensors that already live on the network's device, returning TENSORS (loss, priorities [B]): no host
    synchronisation anywhere, so the whole update can be captured into a HIP graph (`GraphedTrainStep`)."""
    K = action.shape[1]
    mse_v, mse_r = network.mse_loss_for_value, network.mse_loss_for_reward
    pi_logits, value_out, reward_out = unroll(network, state, action)
    value_target = value_scalar.unsqueeze(-1) if mse_v else scalar_to_categorical_probabilities(value_scalar, network.value_support_size)
    reward_target = reward_scalar.unsqueeze(-1) if mse_r else scalar_to_categorical_probabilities(reward_scalar, network.reward_support_size)
    v_loss = loss_func(value_out, value_target, mse_v)
    r_loss = loss_func(reward_out, reward_target, mse_r)
    if mse_v:
        v_loss = v_loss.squeeze(-1)
    if mse_r:
        r_loss = r_loss.squeeze(-1)
    per_sample = (r_loss + v_loss + loss_func(pi_logits, pi_target)).sum(dim=1)  # [B]
    loss = scale_gradient((per_sample * weights.detach()).mean(), 1.0 / K)
    with torch.no_grad():
        v0 = value_out[:, 0]
        v0_scalar = v0.squeeze(-1) if mse_v else logits_to_transformed_expected_value(v0, network.value_support_size).squeeze(-1)
        priorities = (v0_scalar - value_scalar[:, 0]).abs()
    return loss, priorities


def calc_loss(network: MuZeroNet, device: torch.device, transitions: Transition, weights: torch.Tensor) -> Tuple[torch.Tensor, np.ndarray]:
    """The MuZero loss of one batch and the new replay priorities (pipeline.py:541-612).  `transitions` fields may be numpy
    arrays (reference form) or tensors already on `device` (`PrioritizedReplay.sample_tensors`).

    Structure: the scalar targets of all K unroll steps are projected onto their supports at once ([B, K, S]); the network
    is unrolled once into stacked head outputs; each head's per-sample loss is summed over K.  The reported loss is the
    importance-weighted batch mean of (reward + value + policy) -- summed, not averaged, over the K steps -- while its
    GRADIENT is scaled by 1/K (pipeline.py:600).  Priorities are the step-0 value errors in scalar space (:609)."""
    state = _as_tensor(transitions.state, device, torch.float32)           # [B, *state_shape]
    action = _as_tensor(transitions.action, device, torch.long)            # [B, K]
    value_scalar = _as_tensor(transitions.value, device, torch.float32)    # [B, K]
    reward_scalar = _as_tensor(transitions.reward, device, torch.float32)  # [B, K]
    pi_target = _as_tensor(transitions.pi_prob, device, torch.float32)     # [B, K, A]
    loss, priorities = loss_tensors(network, state, action, value_scalar, reward_scalar, pi_target, weights)
    return loss, priorities.cpu().numpy()


# ---------------------------------------------------------------------------------------------------------------
# data-parallel learner: one flat gradient all-reduce per step
# ---------------------------------------------------------------------------------------------------------------
def allreduce_gradients(network: torch.nn.Module, bucket_bytes: int = 64 << 20) -> None:
    """Average the gradients of all ranks (`torch.distributed`, backend "nccl" == RCCL on ROCm, "gloo" on CPU).
    Gradients are packed into flat buckets of up to `bucket_bytes` (default 64 MiB: every reference network fits one
    bucket, i.e. one ring all-reduce per step) so the per-link xGMI ring runs near its bandwidth instead of paying a
    latency per parameter tensor.  No-op without an initialised process group or with a single rank."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    params = [p for p in network.parameters() if p.grad is not None]
    bucket, size = [], 0

    def flush():
        if not bucket:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        off = 0
        for p in bucket:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n

    for p in params:
        nbytes = p.grad.numel() * p.grad.element_size()
        if bucket and size + nbytes > bucket_bytes:
            flush()
            bucket, size = [], 0
        bucket.append(p)
        size += nbytes
    flush()


def train_step(config, network, optimizer, lr_scheduler, device, transitions, weights) -> Tuple[float, np.ndarray]:
    """One learner update (the body of the loop at pipeline.py:238-255): loss, backward, gradient all-reduce across
    learner ranks (if any), optional clipping, Adam step, LR schedule step."""
    weights = _as_tensor(weights, device, torch.float32)
    optimizer.zero_grad()
    loss, priorities = calc_loss(network, device, transitions, weights)
    loss.backward()
    allreduce_gradients(network)
    if config.clip_grad:
        torch.nn.utils.clip_grad_norm_(network.parameters(), config.max_grad_norm)
    optimizer.step()
    lr_scheduler.step()
    return float(loss.detach()), priorities


class GraphedTrainStep:
    """The whole learner update of `train_step` -- unroll forward, loss, backward, gradient clipping, Adam -- captured ONCE as a HIP
    graph and replayed per step.  The MLP learner step is launch-bound: ~400 small kernels for a 128-sample batch, each a few
    microseconds of GPU work behind a Python / dispatcher round trip; replayed as one graph the GPU runs them back to back
    (`tools/learner_bench.py`: ms per step eager vs graphed, and the parity of the two).  Same arithmetic as the eager step: the same
    autograd graph, `torch.optim.Adam(..., capturable=True)` (step count and bias corrections as device tensors), the learning rate a
    device tensor that `MultiStepLR` fills in place, so the schedule is not baked into the capture.

    Requirements: one learner rank (a gradient all-reduce would have to be captured too: the eager `train_step` remains the
    multi-rank path), fixed batch shapes, an optimizer built by `make_capturable_adam`.  `__call__` returns device tensors (loss,
    priorities): nothing synchronises with the host unless the caller reads them."""

    def __init__(self, config, network, optimizer, device, batch_size: int, state_shape, unroll_steps: int, num_actions: int):
        if not all(g.get('capturable', False) for g in optimizer.param_groups):
            raise ValueError('GraphedTrainStep needs an optimizer with capturable=True (learner.make_capturable_adam)')
        self.config, self.network, self.optimizer = config, network, optimizer
        B, K, A = batch_size, unroll_steps, num_actions
        z = lambda *shape, dtype=torch.float32: torch.zeros(*shape, dtype=dtype, device=device)  # noqa: E731
        self.state, self.action = z(B, *state_shape), z(B, K, dtype=torch.long)
        self.value, self.reward, self.pi, self.weights = z(B, K), z(B, K), torch.full((B, K, A), 1.0 / A, device=device), torch.ones(B, device=device)
        # warm-up on a side stream (allocator pools, lazy optimizer state), then restore weights and optimizer state: the warm-up
        # steps must not count as training
        import copy

        net_state = copy.deepcopy(network.state_dict())
        fresh = len(optimizer.state) == 0
        opt_state = None if fresh else copy.deepcopy(optimizer.state_dict())
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):
                self._step_body()
        torch.cuda.current_stream(device).wait_stream(side)
        network.load_state_dict(net_state)
        if fresh:
            # a fresh optimizer goes back to zero moments and step IN PLACE: restoring an empty state would make Adam allocate and
            # zero its state inside the capture, and every replay would then reset the moments
            for st in optimizer.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        else:
            optimizer.load_state_dict(opt_state)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        # CAPTURE BEFORE ANY ACTOR THREAD STARTS (prepare_graphed_step below): a null-stream hipMemcpy / hipMalloc from another thread of
        # the process (a planner's load_state_dict, selfplay_read) synchronises with every blocking stream and invalidates a capture in
        # progress -- in thread-local mode too (measured: hipErrorStreamCaptureInvalidated, tests/test_gpu_learner.py); REPLAYING the
        # graph beside such threads is fine
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            self.loss, self.priorities = self._step_body()

    def _step_body(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss, priorities = loss_tensors(self.network, self.state, self.action, self.value, self.reward, self.pi, self.weights)
        loss.backward()
        if self.config.clip_grad:
            torch.nn.utils.clip_grad_norm_(self.network.parameters(), self.config.max_grad_norm)
        self.optimizer.step()
        return loss.detach(), priorities

    def __call__(self, transitions: Transition, weights) -> Tuple[torch.Tensor, torch.Tensor]:
        dev = self.state.device
        self.state.copy_(_as_tensor(transitions.state, dev, torch.float32))
        self.action.copy_(_as_tensor(transitions.action, dev, torch.long))
        self.value.copy_(_as_tensor(transitions.value, dev, torch.float32))
        self.reward.copy_(_as_tensor(transitions.reward, dev, torch.float32))
        self.pi.copy_(_as_tensor(transitions.pi_prob, dev, torch.float32))
        self.weights.copy_(_as_tensor(weights, dev, torch.float32))
        self.graph.replay()
        # a graph replay rewrites the parameters without touching their torch version counters: tell the module's HIP inference engine
        # (network.inference_engine compares versions) that the weights changed, as hip_learner.HipLearner.apply does
        self.network._mz_weights_epoch = getattr(self.network, '_mz_weights_epoch', 0) + 1
        return self.loss, self.priorities


def make_capturable_adam(network, config, device):
    """`torch.optim.Adam` as the reference builds it (lr_init, weight_decay: classic/run_training.py:94-95) in the form a HIP graph can
    hold: capturable, learning rate as a device tensor (LR schedulers fill it in place)."""
    return torch.optim.Adam(network.parameters(), lr=torch.tensor(float(config.lr_init), device=device), weight_decay=config.weight_decay,
                            capturable=True)


def make_hip_learner(config, network, device, max_batch: Optional[int] = None):
    """The learner step on hand-written gfx950 kernels (muzero_amd.hip_learner.HipLearner, MLP nets) with the optimizer the
    reference's launchers build (classic/run_training.py:94-95: Adam(lr_init, weight_decay) + MultiStepLR(lr_milestones, lr_decay_rate)).
    Hand `hl.optimizer` / `hl.lr_scheduler` to `run_training` in place of the torch objects."""
    from muzero_amd.hip_learner import HipLearner

    return HipLearner(network, device, config.unroll_steps, max_batch or config.batch_size, lr=config.lr_init, weight_decay=config.weight_decay,
                      milestones=config.lr_milestones, gamma=config.lr_decay_rate, clip_grad=bool(config.clip_grad), max_grad_norm=config.max_grad_norm)


def _all_ranks(flag_any: bool, device) -> bool:
    """True if `flag_any` is set on ANY learner rank (a tiny MAX all-reduce); the local value without a process group."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return flag_any
    dev = device if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([1.0 if flag_any else 0.0], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item() > 0)


def _rank() -> int:
    import torch.distributed as dist

    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def prepare_graphed_step(config, network, optimizer, device, state_shape, num_actions, unroll_steps=None, batch_size=None):
    """Build the one-graph update (GraphedTrainStep) for `run_training` AHEAD of the training loop -- in the launcher, before the actor
    threads start: a capture cannot share the process with other threads' null-stream HIP calls (see GraphedTrainStep).  The step is
    left on the optimizer (`optimizer.graphed_step`), where run_training finds it; without it run_training captures lazily at its first
    update, which is only safe when no actor thread of the same process is running yet (the process-per-actor layout)."""
    step = GraphedTrainStep(config, network, optimizer, device, batch_size or config.batch_size, tuple(state_shape),
                            unroll_steps or config.unroll_steps, num_actions)
    optimizer.graphed_step = step
    return step


def run_training(config, network, optimizer, lr_scheduler, device, actor_network, replay: PrioritizedReplay, data_queue, train_steps_counter,
                 checkpoint_dir: str, checkpoint_files: List, stop_event, tag: Optional[str] = None, stop_grace_seconds: float = 10.0) -> None:
    """pipeline.py:170-286: the learner loop that paces the pipeline.  Same arguments; the reference's tensorboard trackers are
    the JSON-lines metrics of `muzero_amd.metrics` (same tag names).

    Data-parallel learners (an initialised `torch.distributed` group, one rank per GPU, each with its own replay shard):
    the decisions that gate the gradient all-reduce are COLLECTIVE -- a step is taken only when every rank's replay is
    warm, and all ranks stop together as soon as any rank sees `stop_event` or reaches `num_training_steps` -- so no rank
    is left waiting in an all-reduce; rank 0 alone writes checkpoints (all ranks hold identical weights) and the other
    ranks pass a barrier before anyone consumes the file."""
    import torch.distributed as dist

    ckpt_prefix = 'train_steps' if not tag else f'{tag}_train_steps'
    network = network.to(device=device)
    network.train()
    rank = _rank()
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    ckpt_dir = Path(checkpoint_dir) if checkpoint_dir else None
    if ckpt_dir is not None and rank == 0:
        ckpt_dir.mkdir(parents=True, exist_ok=True)
    from muzero_amd import metrics as mzm

    metrics = mzm.LearnerMetrics(mzm.run_file(config, 'learner', tag) if rank == 0 else None)
    graphed = getattr(optimizer, 'graphed_step', None)  # prepare_graphed_step(...): captured by the launcher before the actors started
    hip = getattr(optimizer, 'hip_learner', None)  # make_hip_learner(...).optimizer: the update runs on the HIP kernels

    def snapshot():
        return {'network': network.state_dict(), 'optimizer': optimizer.state_dict(), 'lr_scheduler': lr_scheduler.state_dict(),
                'train_steps': train_steps_counter.value}

    def save(name):
        if ckpt_dir is None:
            return None
        path = ckpt_dir / name
        if rank == 0:
            create_checkpoint(snapshot(), path)
        if multi:
            dist.barrier()  # nobody loads a checkpoint that rank 0 is still writing
        return path

    sampler = None  # HIP learner + a replay whose bookkeeping lives on the device: draws and priority updates stay there (replay.DeviceSampler)
    if hip is not None and getattr(replay, '_attached', None) is not None and torch.device(device).type == 'cuda':
        sampler = replay.device_sampler(seed=int(getattr(config, 'seed', 0)) + 7919 * rank)
    if graphed is not None and multi:
        # a captured update has no gradient all-reduce inside: with several learner ranks the weights would drift apart silently
        import warnings

        warnings.warn('run_training: optimizer.graphed_step ignored with %d learner ranks (the one-graph update has no all-reduce); '
                      'using the eager step' % dist.get_world_size())
        graphed = None
    metrics_every = max(1, int(getattr(config, 'metrics_every', 100) or 100))
    loss_acc, loss_n = None, 0
    warm = False
    while True:
        if not warm:  # (replay.size reads the device counter: once the ring is warm it stays warm, and the loop stops asking)
            cold = replay.size < config.min_replay_size or replay.size < config.batch_size
            if _all_ranks(stop_event.is_set() and cold, device):
                return  # stopped before training could start (pipeline.py:232-236), on every rank at once
            if _all_ranks(cold, device):
                time.sleep(0.001)
                continue
            warm = True
        if _all_ranks(train_steps_counter.value >= config.num_training_steps, device):
            break
        transitions = indices = weights = None
        if hip is not None:
            # the kernels gather the batch from the HBM ring by index; gradients of several learner ranks meet in one all-reduce inside step().
            # Nothing below synchronises with the device except the metrics line every 100 updates (and at checkpoints): the reference
            # logs every step (pipeline.py:252-255) because its loss is already on the host; here that read-back would serialise the loop
            if sampler is not None:
                idx_t, w_t, ring = sampler.sample(config.batch_size)
            else:
                indices, is_w, ring = replay.sample_indices(config.batch_size)
                idx_t = torch.from_numpy(indices).to(device)
                w_t = None if replay._alpha == 0 else torch.from_numpy(np.asarray(is_w, np.float32)).to(device)
            loss_t, prio_t = hip.step(ring, idx_t, w_t, config.batch_size)
            if tuple(prio_t.shape) != (config.batch_size,):
                raise RuntimeError(f'Expect priorities has shape ({config.batch_size}, ), got {tuple(prio_t.shape)}')
            # priorities are written on every update, whatever the sampling exponent (pipeline.py:249-252): a saved replay state then
            # carries the reference's priorities even if the exponent changes on a resume
            if sampler is not None:
                sampler.update_priorities(idx_t, prio_t)  # three enqueued kernels, no synchronisation
            else:
                replay.update_priorities(indices, prio_t.cpu().numpy())
            train_steps_counter.value += 1
            # metrics: the reference's loss is on the host after every step; here it stays on the device and the line is written every
            # `config.metrics_every` updates (default 100; 1 = the reference's cadence, one read-back per update) and at checkpoints, carrying
            # the MEAN loss of the updates since the last line (accumulated on the device)
            loss_acc = loss_t.detach().clone() if loss_acc is None else loss_acc.add_(loss_t.detach())
            loss_n += 1
            if train_steps_counter.value % metrics_every == 0 or train_steps_counter.value % config.checkpoint_interval == 0:
                metrics.step(float(loss_acc) / loss_n, lr_scheduler.get_last_lr()[0], train_steps_counter.value)
                loss_acc, loss_n = None, 0
                if sampler is not None:
                    sampler.check_errors()  # what replay.update_priorities / sample would have raised (replay.py:83-84, 106-110)
        else:
            transitions, indices, weights = replay.sample_tensors(config.batch_size)
            if graphed is None and not multi and torch.device(device).type == 'cuda' and all(g.get('capturable', False) for g in optimizer.param_groups):
                # an optimizer built by make_capturable_adam on one learner rank: the update runs as one HIP graph (GraphedTrainStep)
                graphed = GraphedTrainStep(config, network, optimizer, device, config.batch_size, tuple(transitions.state.shape[1:]),
                                           int(transitions.action.shape[1]), int(transitions.pi_prob.shape[2]))
            if graphed is not None:
                loss_t, prio_t = graphed(transitions, weights)
                lr_scheduler.step()
                loss, priorities = float(loss_t), prio_t.cpu().numpy()
            else:
                loss, priorities = train_step(config, network, optimizer, lr_scheduler, device, transitions, weights)
            if priorities is not None:
                if priorities.shape != (config.batch_size,):
                    raise RuntimeError(f'Expect priorities has shape ({config.batch_size}, ), got {priorities.shape}')
                replay.update_priorities(indices, priorities)
            train_steps_counter.value += 1
            metrics.step(loss, lr_scheduler.get_last_lr()[0], train_steps_counter.value)
        del transitions, indices, weights
        if train_steps_counter.value > 1 and train_steps_counter.value % config.checkpoint_interval == 0:
            path = save(f'{ckpt_prefix}_{train_steps_counter.value}')
            if path is not None:
                checkpoint_files.append(path)
            # new values first, THEN the signal: actors in this process see the tensors' versions change, actors in other
            # processes (shared-memory parameters) see the shared `weights_epoch` buffer -- pipeline.run_self_play reloads on either
            if hasattr(actor_network, 'publish_weights'):
                actor_network.publish_weights(network.state_dict())
            else:
                actor_network.load_state_dict(network.state_dict())
            actor_network.eval()
        if config.train_delay is not None and config.train_delay > 0 and train_steps_counter.value > 1:
            time.sleep(config.train_delay)

    stop_event.set()
    time.sleep(stop_grace_seconds)
    data_queue.put('STOP')
    save(f'{ckpt_prefix}_{train_steps_counter.value}_final')
    metrics.close()


def run_data_collector(data_queue, replay: PrioritizedReplay, save_frequency: int = 0, save_dir: Optional[str] = None, tag: Optional[str] = None) -> None:
    """Drains the actors' queue into the replay until the 'STOP' sentinel arrives (the job of pipeline.py:491-538);
    optionally snapshots the replay state every `save_frequency` additions."""
    save_path = Path(save_dir) if save_dir and save_frequency > 0 else None
    if save_path is not None:
        save_path.mkdir(parents=True, exist_ok=True)
    stem = f'{tag}_replay' if tag else 'replay'
    while True:
        try:
            item = data_queue.get()
        except (queue.Empty, EOFError):
            continue
        if isinstance(item, str) and item == 'STOP':
            return
        replay.add(*item)
        if save_path is not None and replay.num_added > 1 and replay.num_added % save_frequency == 0:
            torch.save(replay.get_state(), save_path / f'{stem}_{replay.size}_{int(time.time())}')
