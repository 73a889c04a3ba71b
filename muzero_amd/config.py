"""MuZeroConfig and the four factory configs: API-compatible mirror of muzero/config.py (attribute names, defaults and
temperature schedules are pinned against the reference by tests/test_host_api.py via tests/golden/pipe_cases.npz).

Planner-only knobs are extra keyword arguments with defaults, so existing launcher code keeps working."""
import collections
from typing import Callable, List, Optional

KnownBounds = collections.namedtuple('KnownBounds', ['min', 'max'])


class MuZeroConfig:
    """Attribute bag read by the planner (config.py:22-103)."""

    def __init__(
        self,
        discount: float,
        dirichlet_alpha: float,
        num_simulations: int,
        batch_size: int,
        td_steps: int,
        lr_init: float,
        lr_milestones: List[int],
        visit_softmax_temperature_fn: Callable[[int, int], float],
        known_bounds: Optional[KnownBounds] = None,
        num_training_steps: Optional[int] = int(1000e3),
        checkpoint_interval: Optional[int] = int(1e3),
        num_planes: Optional[int] = 256,
        num_res_blocks: Optional[int] = 16,
        hidden_dim: Optional[int] = 64,
        value_support_size: Optional[int] = 1,
        reward_support_size: Optional[int] = 1,
        train_delay: Optional[float] = 0.0,
        min_replay_size: Optional[int] = int(2e4),
        acc_seq_length: Optional[int] = int(200),
        clip_grad: Optional[bool] = False,
        use_tensorboard: Optional[bool] = False,
        is_board_game: Optional[bool] = False,
        # --- planner-only (not in the reference) ---
        num_envs: int = 1,
        planner_seed: int = 1,
    ) -> None:
        # network architecture
        self.num_planes = num_planes
        self.num_res_blocks = num_res_blocks
        self.value_support_size = value_support_size
        self.reward_support_size = reward_support_size
        self.hidden_dim = hidden_dim
        # self-play
        self.visit_softmax_temperature_fn = visit_softmax_temperature_fn
        self.num_simulations = num_simulations
        self.discount = discount
        self.acc_seq_length = acc_seq_length
        self.root_dirichlet_alpha = dirichlet_alpha
        self.root_exploration_eps = 0.25
        self.pb_c_base = 19652
        self.pb_c_init = 1.25
        self.known_bounds = known_bounds
        # training
        self.num_training_steps = num_training_steps
        self.checkpoint_interval = checkpoint_interval
        self.min_replay_size = min_replay_size
        self.batch_size = batch_size
        self.unroll_steps = 5
        self.td_steps = td_steps
        self.weight_decay = 1e-4
        self.momentum = 0.9
        self.clip_grad = clip_grad
        self.max_grad_norm = 40.0
        self.lr_init = lr_init
        self.lr_decay_rate = 0.1
        self.lr_milestones = lr_milestones
        self.use_tensorboard = use_tensorboard
        self.train_delay = train_delay
        self.is_board_game = is_board_game
        # planner
        self.num_envs = num_envs
        self.planner_seed = planner_seed


def tictactoe_visit_softmax_temperature_fn(env_steps, training_steps):
    return 1.0 if env_steps < 6 else 0.1


def gomoku_visit_softmax_temperature_fn(env_steps, training_steps):
    return 1.0 if env_steps < 30 else 0.1


def _by_training_steps(training_steps, first, second):
    if training_steps < first:
        return 1.0
    return 0.5 if training_steps < second else 0.25


def classic_visit_softmax_temperature_fn(env_steps, training_steps):
    return _by_training_steps(training_steps, 30000, 60000)


def atari_visit_softmax_temperature_fn(env_steps, training_steps):
    return _by_training_steps(training_steps, 500e3, 1000e3)


def make_tictactoe_config(num_training_steps=100000, batch_size=128, min_replay_size=10000, use_mlp_net=True, use_tensorboard=True,
                          clip_grad=False) -> MuZeroConfig:
    return MuZeroConfig(
        discount=1.0, dirichlet_alpha=0.25, num_simulations=25, batch_size=batch_size, td_steps=0, lr_init=0.002, lr_milestones=[20000],
        visit_softmax_temperature_fn=tictactoe_visit_softmax_temperature_fn, known_bounds=KnownBounds(-1, 1),
        num_training_steps=num_training_steps, num_planes=256 if use_mlp_net else 16, num_res_blocks=0 if use_mlp_net else 2,
        hidden_dim=64 if use_mlp_net else 0, min_replay_size=min_replay_size, checkpoint_interval=500, acc_seq_length=9999, train_delay=0.0,
        clip_grad=clip_grad, use_tensorboard=use_tensorboard, is_board_game=True,
    )


def make_gomoku_config(num_training_steps=1000000, batch_size=128, min_replay_size=10000, use_tensorboard=True, clip_grad=False) -> MuZeroConfig:
    return MuZeroConfig(
        discount=1.0, dirichlet_alpha=0.03, num_simulations=200, batch_size=batch_size, td_steps=0, lr_init=0.002,
        lr_milestones=[200e3, 400e3], visit_softmax_temperature_fn=gomoku_visit_softmax_temperature_fn, known_bounds=KnownBounds(-1, 1),
        num_training_steps=num_training_steps, num_planes=128, num_res_blocks=8, hidden_dim=0, min_replay_size=min_replay_size,
        acc_seq_length=9999, train_delay=0.0, clip_grad=clip_grad, use_tensorboard=use_tensorboard, is_board_game=True,
    )


def make_classic_config(num_training_steps=100000, batch_size=256, min_replay_size=10000, use_tensorboard=True, clip_grad=False) -> MuZeroConfig:
    return MuZeroConfig(
        discount=0.997, dirichlet_alpha=0.25, num_simulations=50, batch_size=batch_size, td_steps=10, lr_init=0.005, lr_milestones=[20000],
        visit_softmax_temperature_fn=classic_visit_softmax_temperature_fn, num_training_steps=num_training_steps, num_planes=512,
        num_res_blocks=0, hidden_dim=64, value_support_size=31, reward_support_size=31, min_replay_size=min_replay_size,
        checkpoint_interval=200, acc_seq_length=9999, train_delay=0.0, clip_grad=clip_grad, use_tensorboard=use_tensorboard,
        is_board_game=False,
    )


def make_atari_config(num_training_steps=int(10e6), batch_size=128, min_replay_size=10000, use_tensorboard=True, clip_grad=False) -> MuZeroConfig:
    return MuZeroConfig(
        discount=0.997, dirichlet_alpha=0.25, num_simulations=30, batch_size=batch_size, td_steps=10, lr_init=0.05,
        lr_milestones=[100e3, 200e3], visit_softmax_temperature_fn=atari_visit_softmax_temperature_fn,
        num_training_steps=num_training_steps, num_planes=128, num_res_blocks=8, hidden_dim=0, value_support_size=61,
        reward_support_size=61, min_replay_size=min_replay_size, acc_seq_length=200, train_delay=0.0, clip_grad=clip_grad,
        use_tensorboard=use_tensorboard, is_board_game=False,
    )
