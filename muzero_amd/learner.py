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
    """util.py:48-59: scalar -> two adjacent support bins holding its linear interpolation weights."""
    scalar = torch.clamp(scalar, min_value, max_value)
    scalar_bin = (scalar - min_value) / (max_value - min_value) * (num_bins - 1)
    lower, upper = torch.floor(scalar_bin), torch.ceil(scalar_bin)
    lower_value = (lower / (num_bins - 1.0)) * (max_value - min_value) + min_value
    upper_value = (upper / (num_bins - 1.0)) * (max_value - min_value) + min_value
    p_lower = (upper_value - scalar) / (upper_value - lower_value + 1e-5)
    p_upper = 1 - p_lower
    lower_one_hot = F.one_hot(lower.long(), num_bins) * torch.unsqueeze(p_lower, -1)
    upper_one_hot = F.one_hot(upper.long(), num_bins) * torch.unsqueeze(p_upper, -1)
    return lower_one_hot + upper_one_hot


def scalar_to_categorical_probabilities(x: torch.Tensor, support_size: int) -> torch.Tensor:
    """util.py:96-116: signed_hyperbolic, then projection onto the integer support [-(S-1)/2, (S-1)/2]."""
    x = signed_hyperbolic(x)
    max_value = (support_size - 1) // 2
    return transform_to_2hot(x, -max_value, max_value, support_size)


def loss_func(prediction: torch.Tensor, target: torch.Tensor, mse: bool = False) -> torch.Tensor:
    """pipeline.py:615-629: per-sample MSE (scalar heads) or soft-target cross entropy (categorical heads, policy)."""
    assert prediction.shape == target.shape
    if not mse:
        assert len(prediction.shape) == 2
    if mse:
        return F.mse_loss(prediction, target, reduction='none')
    return F.cross_entropy(prediction, target, reduction='none')


def _as_tensor(x, device, dtype):
    t = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x))
    return t.to(device=device, dtype=dtype, non_blocking=True)


def calc_loss(network: MuZeroNet, device: torch.device, transitions: Transition, weights: torch.Tensor) -> Tuple[torch.Tensor, np.ndarray]:
    """pipeline.py:541-612.  `transitions` fields may be numpy arrays (reference form) or tensors already on `device`
    (`PrioritizedReplay.sample_tensors`)."""
    state = _as_tensor(transitions.state, device, torch.float32)                   # [B, *state_shape]
    action = _as_tensor(transitions.action, device, torch.long)                    # [B, T]
    target_value_scalar = _as_tensor(transitions.value, device, torch.float32)     # [B, T]
    target_reward_scalar = _as_tensor(transitions.reward, device, torch.float32)   # [B, T]
    target_pi_prob = _as_tensor(transitions.pi_prob, device, torch.float32)        # [B, T, A]

    target_value = target_value_scalar if network.mse_loss_for_value else scalar_to_categorical_probabilities(
        target_value_scalar, network.value_support_size)
    target_reward = target_reward_scalar if network.mse_loss_for_reward else scalar_to_categorical_probabilities(
        target_reward_scalar, network.reward_support_size)

    B, T = action.shape
    reward_loss, value_loss, policy_loss = (0, 0, 0)
    loss_scale = 1.0 / T
    pred_values = []

    hidden_state = network.represent(state)
    for t in range(T):  # unroll K steps
        pred_pi_logits, pred_value = network.prediction(hidden_state)
        hidden_state, pred_reward = network.dynamics(hidden_state, action[:, t].unsqueeze(1))
        hidden_state.register_hook(lambda grad: grad * 0.5)
        value_loss += loss_func(pred_value.squeeze(), target_value[:, t], network.mse_loss_for_value)
        reward_loss += loss_func(pred_reward.squeeze(), target_reward[:, t], network.mse_loss_for_reward)
        policy_loss += loss_func(pred_pi_logits, target_pi_prob[:, t])
        pred_values.append(pred_value.detach())

    loss = reward_loss + value_loss + policy_loss
    loss = torch.mean(loss * weights.detach())
    loss.register_hook(lambda grad: grad * loss_scale)

    with torch.no_grad():
        pred_values = torch.stack(pred_values, dim=1)
        if network.mse_loss_for_value:
            pred_values_scalar = pred_values.squeeze(-1)
        else:
            pred_values_scalar = logits_to_transformed_expected_value(pred_values, network.value_support_size).squeeze(-1)
        priorities = (pred_values_scalar[:, 0] - target_value_scalar[:, 0]).abs().cpu().numpy()
    return loss, priorities


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


def run_training(config, network, optimizer, lr_scheduler, device, actor_network, replay: PrioritizedReplay, data_queue, train_steps_counter,
                 checkpoint_dir: str, checkpoint_files: List, stop_event, tag: Optional[str] = None, stop_grace_seconds: float = 10.0) -> None:
    """pipeline.py:170-286: the learner loop that paces the pipeline.  Same arguments; trackers/tensorboard are out of
    scope.  With an initialised `torch.distributed` group every rank runs this loop on its own replay shard and the
    gradients are averaged each step, so all ranks hold identical weights."""
    ckpt_prefix = 'train_steps' if not tag else f'{tag}_train_steps'
    network = network.to(device=device)
    network.train()
    ckpt_dir = Path(checkpoint_dir) if checkpoint_dir else None
    if ckpt_dir is not None and not ckpt_dir.exists():
        ckpt_dir.mkdir(parents=True, exist_ok=True)

    def get_state_to_save():
        return {'network': network.state_dict(), 'optimizer': optimizer.state_dict(), 'lr_scheduler': lr_scheduler.state_dict(),
                'train_steps': train_steps_counter.value}

    while True:
        if replay.size < config.min_replay_size or replay.size < config.batch_size:
            time.sleep(0.001)
            if stop_event.is_set():
                return
            continue
        if train_steps_counter.value >= config.num_training_steps:
            break
        transitions, indices, weights = replay.sample_tensors(config.batch_size)
        loss, priorities = train_step(config, network, optimizer, lr_scheduler, device, transitions, weights)
        if priorities is not None:
            if priorities.shape != (config.batch_size,):
                raise RuntimeError(f'Expect priorities has shape ({config.batch_size}, ), got {priorities.shape}')
            replay.update_priorities(indices, priorities)
        train_steps_counter.value += 1
        del transitions, indices, weights
        if train_steps_counter.value > 1 and train_steps_counter.value % config.checkpoint_interval == 0:
            if ckpt_dir is not None:
                ckpt_file = ckpt_dir / f'{ckpt_prefix}_{train_steps_counter.value}'
                create_checkpoint(get_state_to_save(), ckpt_file)
                checkpoint_files.append(ckpt_file)
            actor_network.load_state_dict(network.state_dict())  # the planner reloads on the parameter-version bump
            actor_network.eval()
        if config.train_delay is not None and config.train_delay > 0 and train_steps_counter.value > 1:
            time.sleep(config.train_delay)

    stop_event.set()
    time.sleep(stop_grace_seconds)
    data_queue.put('STOP')
    if ckpt_dir is not None:
        create_checkpoint(get_state_to_save(), ckpt_dir / f'{ckpt_prefix}_{train_steps_counter.value}_final')


def run_data_collector(data_queue, replay: PrioritizedReplay, save_frequency: int = 0, save_dir: Optional[str] = None, tag: Optional[str] = None) -> None:
    """pipeline.py:491-538: moves `(Transition, priority)` items from the actors' queue into the replay until 'STOP'."""
    prefix = 'replay' if not tag else f'{tag}_replay'
    save_path = Path(save_dir) if save_dir else None
    if save_path is not None and not save_path.exists():
        save_path.mkdir(parents=True, exist_ok=True)
    should_save = save_path is not None and save_frequency > 0
    while True:
        try:
            item = data_queue.get()
            if isinstance(item, str) and item == 'STOP':
                break
            transition, priority = item
            replay.add(transition, priority)
            if should_save and replay.num_added > 1 and replay.num_added % save_frequency == 0:
                torch.save(replay.get_state(), save_path / f'{prefix}_{replay.size}_{int(time.time())}')
        except queue.Empty:
            pass
        except EOFError:
            pass
