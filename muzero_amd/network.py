"""MuZero networks: host-side mirror of the reference model API (muzero/network.py).

Two faces:

* The torch ``nn.Module`` tree exists for *parameters*: its ``state_dict()`` has exactly the
  reference's key names, order and shapes (network.py:140-574), so checkpoints written by either
  implementation load into the other (``{'network', 'optimizer', 'lr_scheduler', 'train_steps'}``,
  pipeline.py:224-230), and the tensor API ``represent / dynamics / prediction`` used by the learner's
  autograd path (pipeline.py:541-629) is plain PyTorch-ROCm.

* ``initial_inference`` / ``recurrent_inference`` (network.py:62-111) -- the calls on the planning hot
  path -- do not run torch ops at all: they hand the parameters to the HIP planner
  (``libmzplanner_hip.so``, C-ABI in include/mzplanner.h) and execute its fused inference kernels.
  They raise if the planner library or a GPU is not available: there is no CPU fallback.
"""
import math
from typing import NamedTuple, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


class NetworkOutputs(NamedTuple):
    # same field names/order as network.py:25-30
    hidden_state: torch.Tensor
    reward: torch.Tensor
    pi_probs: torch.Tensor
    value: torch.Tensor


def normalize_hidden_state(hidden_state: torch.Tensor) -> torch.Tensor:
    """util.py:31-36 -- min/max over dim=1, +1e-8 in the denominator."""
    lo = hidden_state.amin(dim=1, keepdim=True)
    hi = hidden_state.amax(dim=1, keepdim=True)
    return (hidden_state - lo) / (hi - lo + 1e-8)


def signed_hyperbolic(x: torch.Tensor, eps: float = 1e-3) -> torch.Tensor:
    """util.py:20-22"""
    return torch.sign(x) * (torch.sqrt(torch.abs(x) + 1) - 1) + eps * x


def signed_parabolic(x: torch.Tensor, eps: float = 1e-3) -> torch.Tensor:
    """util.py:25-28"""
    z = torch.sqrt(1 + 4 * eps * (eps + 1 + torch.abs(x))) / 2 / eps - 1 / 2 / eps
    return torch.sign(x) * (torch.square(z) - 1)


def logits_to_transformed_expected_value(logits: torch.Tensor, support_size: int) -> torch.Tensor:
    """util.py:70-93 (support built on the logits' device, unlike util.py:64 which pins it to the CPU)."""
    half = (support_size - 1) // 2
    support = torch.linspace(-half, half, support_size, device=logits.device, dtype=logits.dtype)
    expected = (torch.softmax(logits, dim=-1) * support).sum(dim=-1, keepdim=True)
    return signed_parabolic(expected)


def _two_layer(n_in: int, n_mid: int, n_out: int) -> nn.Sequential:
    return nn.Sequential(nn.Linear(n_in, n_mid), nn.ReLU(), nn.Linear(n_mid, n_out))


def _conv3x3(c_in: int, c_out: int, stride: int = 1) -> nn.Conv2d:
    return nn.Conv2d(c_in, c_out, kernel_size=3, stride=stride, padding=1, bias=False)


def _conv_bn_relu(c_in: int, c_out: int) -> nn.Sequential:
    return nn.Sequential(_conv3x3(c_in, c_out), nn.BatchNorm2d(c_out), nn.ReLU())


def _plane_head(c_in: int, planes: int, hw: int, n_out: int) -> nn.Sequential:
    # indices 0,1,4 carry parameters -> keys '<head>.0.weight', '<head>.1.*', '<head>.4.{weight,bias}'
    return nn.Sequential(
        nn.Conv2d(c_in, planes, kernel_size=1, stride=1, bias=False), nn.BatchNorm2d(planes), nn.ReLU(), nn.Flatten(),
        nn.Linear(planes * hw, n_out),
    )


def initialize_weights(net: nn.Module) -> None:
    """network.py:33-45 -- kaiming-normal for Conv2d/Linear weights, zero biases."""
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.kaiming_normal_(m.weight, nonlinearity='relu')
            if m.bias is not None:
                nn.init.zeros_(m.bias)


class ResNetBlock(nn.Module):
    """network.py:273-299"""

    def __init__(self, num_planes: int) -> None:
        super().__init__()
        self.conv_block1 = _conv_bn_relu(num_planes, num_planes)
        self.conv_block2 = nn.Sequential(_conv3x3(num_planes, num_planes), nn.BatchNorm2d(num_planes))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.relu(self.conv_block2(self.conv_block1(x)) + x)


def _tower(num_planes: int, n: int) -> nn.Sequential:
    return nn.Sequential(*[ResNetBlock(num_planes) for _ in range(n)])


class RepresentationMLPNet(nn.Module):
    def __init__(self, input_size: int, num_planes: int, hidden_dim: int) -> None:
        super().__init__()
        self.net = _two_layer(input_size, num_planes, hidden_dim)

    def forward(self, x):
        return self.net(x.reshape(x.shape[0], -1))


class DynamicsMLPNet(nn.Module):
    def __init__(self, num_actions: int, num_planes: int, hidden_dim: int, support_size: int) -> None:
        super().__init__()
        self.num_actions = num_actions
        self.transition_net = _two_layer(hidden_dim + num_actions, num_planes, hidden_dim)
        self.reward_net = _two_layer(hidden_dim, num_planes, support_size)

    def forward(self, hidden_state, action):
        onehot = F.one_hot(action.reshape(-1).long(), self.num_actions).to(hidden_state.dtype)
        nxt = self.transition_net(torch.cat([hidden_state, onehot], dim=1))
        return nxt, self.reward_net(nxt)  # reward from the un-normalised state (network.py:195-196)


class PredictionMLPNet(nn.Module):
    def __init__(self, num_actions: int, num_planes: int, hidden_dim: int, support_size: int) -> None:
        super().__init__()
        self.policy_net = _two_layer(hidden_dim, num_planes, num_actions)
        self.value_net = _two_layer(hidden_dim, num_planes, support_size)

    def forward(self, hidden_state):
        return self.policy_net(hidden_state), self.value_net(hidden_state)


class RepresentationConvAtariNet(nn.Module):
    """network.py:312-353: 96x96 -> 48 -> 24 -> 12 -> 6"""

    def __init__(self, input_shape: Tuple, num_planes: int) -> None:
        super().__init__()
        c = input_shape[0]
        self.conv_1 = _conv3x3(c, 128, stride=2)
        self.res_blocks_1 = _tower(128, 2)
        self.conv_2 = _conv3x3(128, num_planes, stride=2)
        self.res_blocks_2 = _tower(num_planes, 2)
        self.avg_pool_1 = nn.AvgPool2d(kernel_size=3, stride=2, padding=1)
        self.res_blocks_3 = _tower(num_planes, 2)
        self.avg_pool_2 = nn.AvgPool2d(kernel_size=3, stride=2, padding=1)

    def forward(self, x):
        x = self.res_blocks_1(F.relu(self.conv_1(x)))
        x = self.res_blocks_2(F.relu(self.conv_2(x)))
        x = self.res_blocks_3(self.avg_pool_1(x))
        return self.avg_pool_2(x)


class RepresentationConvNet(nn.Module):
    def __init__(self, input_shape: Tuple, num_planes: int, num_res_block: int) -> None:
        super().__init__()
        self.conv_block = _conv_bn_relu(input_shape[0], num_planes)
        self.res_blocks = _tower(num_planes, num_res_block)

    def forward(self, x):
        return self.res_blocks(self.conv_block(x))


def reference_action_planes(action: torch.Tensor, num_actions: int, h: int, w: int, dtype) -> torch.Tensor:
    """Action planes exactly as the reference builds them (network.py:440-444): with action of shape [B,1] the
    one-hot is repeated h*w times along dim 1 and reshaped, so element f = c*h*w + y*w + x of the [A,h,w] block
    is 1 iff f % A == action (not a constant plane per action).  Closed form, no intermediate [B,h*w,A] tensor."""
    f = torch.arange(num_actions * h * w, device=action.device) % num_actions
    return (f[None, :] == action.reshape(-1, 1).long()).to(dtype).reshape(-1, num_actions, h, w)


class DynamicsConvNet(nn.Module):
    def __init__(self, input_shape: Tuple, num_actions: int, num_res_block: int, num_planes: int, support_size: int) -> None:
        super().__init__()
        self.num_actions = num_actions
        c, h, w = input_shape
        self.conv_block = _conv_bn_relu(c, num_planes)
        self.res_blocks = _tower(num_planes, num_res_block)
        self.reward_head = _plane_head(num_planes, 1, h * w, support_size)

    def forward(self, hidden_state, action):
        _, _, h, w = hidden_state.shape
        planes = reference_action_planes(action, self.num_actions, h, w, hidden_state.dtype)
        nxt = self.res_blocks(self.conv_block(torch.cat([hidden_state, planes], dim=1)))
        return nxt, self.reward_head(nxt)


class PredictionConvNet(nn.Module):
    def __init__(self, input_shape: Tuple, num_actions: int, num_res_block: int, num_planes: int, support_size: int) -> None:
        super().__init__()
        _, h, w = input_shape
        self.res_blocks = _tower(num_planes, num_res_block)
        self.policy_net = _plane_head(num_planes, 2, h * w, num_actions)
        self.value_net = _plane_head(num_planes, 1, h * w, support_size)

    def forward(self, hidden_state):
        feat = self.res_blocks(hidden_state)
        return self.policy_net(feat), self.value_net(feat)


class MuZeroNet(nn.Module):
    """Base class (network.py:48-134).  Inference on the planning path goes through the HIP planner."""

    kind = None  # 'mlp' | 'board' | 'atari'

    def __init__(self, num_actions: int, value_support_size: int = 31, reward_support_size: int = 31) -> None:
        super().__init__()
        self.num_actions = num_actions
        self.value_support_size = value_support_size
        self.reward_support_size = reward_support_size
        self._engine = None
        self._engine_version = None
        # Published-weights epoch: bumped by `publish_weights` AFTER the new values are in place.  A non-persistent buffer, so
        # the state_dict / checkpoint layout stays the reference's, and `share_memory()` shares it with the parameters: it is
        # the refresh signal that crosses process boundaries (tensor._version does not).
        self.register_buffer('weights_epoch', torch.zeros(1, dtype=torch.int64), persistent=False)

    def publish_weights(self, state_dict) -> None:
        """What the learner does for its actors at a checkpoint boundary (pipeline.py:261-267): new values first, signal after."""
        self.load_state_dict(state_dict)
        self.weights_epoch.add_(1)

    # --- tensor API (learner side, autograd) ------------------------------------------------
    def represent(self, x: torch.Tensor) -> torch.Tensor:
        return normalize_hidden_state(self.represent_net(x))

    def dynamics(self, hidden_state: torch.Tensor, action: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        nxt, reward_logits = self.dynamics_net(hidden_state, action)
        return normalize_hidden_state(nxt), reward_logits

    def prediction(self, hidden_state: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.prediction_net(hidden_state)

    @property
    def mse_loss_for_value(self) -> bool:
        return self.value_support_size == 1

    @property
    def mse_loss_for_reward(self) -> bool:
        return self.reward_support_size == 1

    # --- planner hand-off -------------------------------------------------------------------
    def planner_spec(self) -> dict:
        raise NotImplementedError

    def __getstate__(self):
        # the bound engine is a handle into libmzplanner_hip.so: it does not travel with a pickled / deep-copied module (the reference's
        # launchers hand `actor_network` to spawned actor processes, */run_training.py); the copy binds its own engine on first use
        state = dict(super().__getstate__())
        state['_engine'] = None
        state['_engine_version'] = None
        return state

    def _weights_version(self):
        # (+ an epoch that writers outside torch bump: hip_learner.HipLearner updates the parameters' storage from its own kernels)
        return tuple(p._version for p in self.parameters()) + tuple(b._version for b in self.buffers()) + (getattr(self, '_mz_weights_epoch', 0),)

    def inference_engine(self, device=None):
        """The HIP inference engine bound to this module's current parameters (rebuilt when they change)."""
        from muzero_amd import planner as _planner  # deferred: loads libmzplanner_hip.so, raises if unavailable

        ver = self._weights_version()
        if self._engine is None or self._engine_version != ver:
            if self._engine is None:
                self._engine = _planner.InferenceEngine(self.planner_spec(), device=device)
            self._engine.load_state_dict(self.state_dict())
            self._engine_version = ver
        return self._engine

    def _shape_hidden(self, flat):
        """hidden state without the batch dimension, as the reference returns it: [H] for MLP nets, [C, h, w] for conv nets"""
        shape = getattr(self, 'hidden_shape', None)
        return flat if shape is None else flat.reshape(shape)

    @torch.no_grad()
    def initial_inference(self, x: torch.Tensor) -> NetworkOutputs:
        """network.py:62-84: batch of one in, numpy / python scalars out."""
        eng = self.inference_engine(x.device)
        hidden, pi, value = eng.initial_inference(x.detach().to(torch.float32).cpu().numpy())
        return NetworkOutputs(hidden_state=self._shape_hidden(hidden[0]), reward=0.0, pi_probs=pi[0], value=float(value[0]))

    @torch.no_grad()
    def recurrent_inference(self, hidden_state: torch.Tensor, action: torch.Tensor) -> NetworkOutputs:
        """network.py:86-111"""
        eng = self.inference_engine(hidden_state.device)
        hidden, reward, pi, value = eng.recurrent_inference(
            hidden_state.detach().to(torch.float32).cpu().numpy(), action.detach().cpu().numpy().reshape(-1).astype(np.int32)
        )
        return NetworkOutputs(hidden_state=self._shape_hidden(hidden[0]), reward=float(reward[0]), pi_probs=pi[0], value=float(value[0]))


class MuZeroMLPNet(MuZeroNet):
    """network.py:236-267"""

    kind = 'mlp'

    def __init__(self, input_shape: Tuple, num_actions: int, num_planes: int = 256, value_support_size: int = 31,
                 reward_support_size: int = 31, hidden_dim: int = 64) -> None:
        super().__init__(num_actions, value_support_size, reward_support_size)
        self.input_shape = tuple(input_shape)
        self.num_planes = num_planes
        self.hidden_dim = hidden_dim
        self.represent_net = RepresentationMLPNet(math.prod(input_shape), num_planes, hidden_dim)
        self.dynamics_net = DynamicsMLPNet(num_actions, num_planes, hidden_dim, reward_support_size)
        self.prediction_net = PredictionMLPNet(num_actions, num_planes, hidden_dim, value_support_size)

    def planner_spec(self) -> dict:
        return dict(kind='mlp', input_shape=self.input_shape, num_actions=self.num_actions, num_planes=self.num_planes,
                    hidden_dim=self.hidden_dim, num_res_blocks=0, value_support_size=self.value_support_size,
                    reward_support_size=self.reward_support_size)


class MuZeroAtariNet(MuZeroNet):
    """network.py:501-537 (hidden state is always [num_planes, 6, 6])"""

    kind = 'atari'

    def __init__(self, input_shape: tuple, num_actions: int, num_res_blocks: int = 16, num_planes: int = 256,
                 value_support_size: int = 601, reward_support_size: int = 601) -> None:
        super().__init__(num_actions, value_support_size, reward_support_size)
        self.input_shape = tuple(input_shape)
        self.num_planes = num_planes
        self.num_res_blocks = num_res_blocks
        self.represent_net = RepresentationConvAtariNet(input_shape, num_planes)
        self.dynamics_net = DynamicsConvNet((num_planes + num_actions, 6, 6), num_actions, num_res_blocks, num_planes, reward_support_size)
        self.prediction_net = PredictionConvNet((num_planes, 6, 6), num_actions, num_res_blocks, num_planes, value_support_size)
        self.hidden_shape = (num_planes, 6, 6)
        initialize_weights(self)

    def planner_spec(self) -> dict:
        return dict(kind='atari', input_shape=self.input_shape, num_actions=self.num_actions, num_planes=self.num_planes,
                    hidden_dim=0, num_res_blocks=self.num_res_blocks, value_support_size=self.value_support_size,
                    reward_support_size=self.reward_support_size)


class MuZeroBoardGameNet(MuZeroNet):
    """network.py:540-574 (MSE heads: support size 1, no tanh on the value)"""

    kind = 'board'

    def __init__(self, input_shape: tuple, num_actions: int, num_res_blocks: int = 16, num_planes: int = 256) -> None:
        super().__init__(num_actions, 1, 1)
        self.input_shape = tuple(input_shape)
        self.num_planes = num_planes
        self.num_res_blocks = num_res_blocks
        _, h, w = input_shape
        self.represent_net = RepresentationConvNet(input_shape, num_planes, num_res_blocks)
        self.dynamics_net = DynamicsConvNet((num_planes + num_actions, h, w), num_actions, num_res_blocks, num_planes, 1)
        self.prediction_net = PredictionConvNet((num_planes, h, w), num_actions, num_res_blocks, num_planes, 1)
        self.hidden_shape = (num_planes, h, w)
        initialize_weights(self)

    def planner_spec(self) -> dict:
        return dict(kind='board', input_shape=self.input_shape, num_actions=self.num_actions, num_planes=self.num_planes,
                    hidden_dim=0, num_res_blocks=self.num_res_blocks, value_support_size=1, reward_support_size=1)
