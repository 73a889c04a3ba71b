"""Host-side board games for evaluators and tools: `BoardGameEnv` semantics of the reference (`games/env.py:40-365`)
with the win test of `games/tictactoe.py:33-77` / `games/gomoku.py:72-116` (n in a row through the last move only).
Self-play does not use this class -- its environments live on the device (`csrc/mz_env.h`, same rules)."""
from typing import Optional, Tuple

import numpy as np


class BoardGameEnv:
    """N x N board, players 1 (black, moves first) and 2 (white); action N*N resigns (`enable_resign`)."""

    def __init__(self, board_size: int = 15, stack_history: int = 4, num_to_win: int = 5, black_player_id: int = 1, white_player_id: int = 2,
                 enable_resign: bool = True, name: str = '') -> None:
        assert black_player_id != white_player_id != 0, 'player ids can not be the same, and can not be zero'
        self.name = name
        self.board_size, self.stack_history, self.num_to_win = board_size, stack_history, num_to_win
        self.black_player_id, self.white_player_id = black_player_id, white_player_id
        self.black_color, self.white_color = 1, 2
        self.num_actions = board_size ** 2 + 1 if enable_resign else board_size ** 2
        self.resign_action: Optional[int] = self.num_actions - 1 if enable_resign else None
        self.observation_shape = (stack_history * 2 + 1, board_size, board_size)
        self.reset()

    def reset(self, **kwargs) -> np.ndarray:
        n = self.board_size
        self.board = np.zeros((n, n), dtype=np.int8)
        self.actions_mask = np.ones(self.num_actions, dtype=np.bool_)
        self.current_player = self.black_player_id
        self.steps = 0
        self.winner: Optional[int] = None
        self.last_actions = {self.black_player_id: None, self.white_player_id: None}
        # own-stone history planes per player, most recent first (games/env.py:294-310)
        self.feature_planes = {p: np.zeros((self.stack_history, n, n), dtype=np.int8) for p in (self.black_player_id, self.white_player_id)}
        return self.observation()

    # ---- properties (games/env.py:312-365) ----
    @property
    def opponent_player(self) -> int:
        return self.white_player_id if self.current_player == self.black_player_id else self.black_player_id

    @property
    def current_player_color(self) -> int:
        return self.black_color if self.current_player == self.black_player_id else self.white_color

    @property
    def is_board_full(self) -> bool:
        return bool(np.all(self.board != 0))

    @property
    def is_game_over(self) -> bool:
        return self.winner is not None or self.is_board_full

    @property
    def loser(self) -> Optional[int]:
        if self.winner is None:
            return None
        return self.white_player_id if self.winner == self.black_player_id else self.black_player_id

    def action_to_coords(self, action: int) -> Tuple[int, int]:
        return action // self.board_size, action % self.board_size

    def coords_to_action(self, coords: Tuple[int, int]) -> int:
        return coords[0] * self.board_size + coords[1]

    def is_action_valid(self, action: int) -> bool:
        return 0 <= action < self.num_actions and bool(self.actions_mask[action])

    # ---- dynamics ----
    def step(self, action: int):
        """games/env.py:117-154: returns (observation, reward, done, info); the player is not switched on the final move."""
        if not 0 <= action <= self.num_actions - 1:
            raise ValueError(f'Invalid action. Expect action to be in range [0, {self.num_actions}], got {action}')
        if not self.actions_mask[action]:
            raise ValueError(f'Invalid action. The action {action} has alread been taken.')
        if self.is_game_over:
            raise RuntimeError('Game is over, call reset before using step method.')
        action = int(action)
        reward = 0.0
        self.actions_mask[action] = False
        self.last_actions[self.current_player] = action
        if action == self.resign_action:
            reward = -1.0
            self.winner = self.opponent_player
        else:
            r, c = self.action_to_coords(action)
            self.board[r, c] = self.current_player_color
            planes = self.feature_planes[self.current_player]
            planes[1:] = planes[:-1].copy()
            planes[0] = (self.board == self.current_player_color)
            if self.is_current_player_won():
                reward = 1.0
                self.winner = self.current_player
        done = self.is_game_over
        if not done:
            self.current_player = self.opponent_player
        self.steps += 1
        return self.observation(), reward, done, {}

    def is_current_player_won(self) -> bool:
        """Lines through the last move only; needs at least 2 * (num_to_win - 1) earlier plies (steps not yet incremented)."""
        if self.steps < (self.num_to_win - 1) * 2:
            return False
        last = self.last_actions[self.current_player]
        if last is None or last == self.resign_action:
            return False
        r0, c0 = self.action_to_coords(last)
        colour, n = self.current_player_color, self.board_size
        for dr, dc in ((0, 1), (1, 0), (1, 1), (-1, 1)):
            count = 1
            for sgn in (1, -1):
                r, c = r0 + sgn * dr, c0 + sgn * dc
                while 0 <= r < n and 0 <= c < n and self.board[r, c] == colour:
                    count += 1
                    r, c = r + sgn * dr, c + sgn * dc
            if count >= self.num_to_win:
                return True
        return False

    def observation(self) -> np.ndarray:
        """[X_t, Y_t, X_t-1, Y_t-1, ..., C] from the side to move, int8 (games/env.py:242-271)."""
        me, opp = self.current_player, self.opponent_player
        n = self.board_size
        out = np.zeros(self.observation_shape, dtype=np.int8)
        out[0:2 * self.stack_history:2] = self.feature_planes[me]
        out[1:2 * self.stack_history:2] = self.feature_planes[opp]
        out[-1] = 1 if me == self.black_player_id else 0
        return out.reshape(self.observation_shape) if n else out


class TicTacToeEnv(BoardGameEnv):
    """games/tictactoe.py:25-31"""

    def __init__(self, stack_history: int = 4) -> None:
        super().__init__(board_size=3, stack_history=stack_history, num_to_win=3, name='TicTacToe')


class GomokuEnv(BoardGameEnv):
    """games/gomoku.py:25-70"""

    def __init__(self, board_size: int = 15, stack_history: int = 4, num_to_win: int = 5) -> None:
        super().__init__(board_size=board_size, stack_history=stack_history, num_to_win=num_to_win, name='Gomoku')


class StackFrameAndAction:
    """`gym_env.py:271-353`: stacks the last `stack_history` observations, newest first, each with the action that led to it
    as a bias value (action + 1) / num_actions (reset fills every slot with the first observation and action 0).  Vector
    observations [D] become [stack, D + 1]; channel-first images [C, H, W] become [stack * (C + 1), H, W] with the action
    broadcast to a plane.  `env` needs reset() -> obs, step(a) -> (obs, reward, done, info), `observation_shape` and
    `num_actions`."""

    def __init__(self, env, stack_history: int, is_obs_image: bool = False) -> None:
        self.env, self.stack_history, self.is_obs_image = env, stack_history, is_obs_image
        self.num_actions = env.num_actions
        shp = tuple(env.observation_shape)
        self.old_obs_shape = shp[1:] if is_obs_image else shp
        self.observation_shape = (stack_history * (shp[0] + 1),) + shp[1:] if is_obs_image else (stack_history, shp[0] + 1)
        self._obs, self._act = [], []

    def __getattr__(self, name):
        return getattr(self.env, name)

    def _plane(self, action) -> np.ndarray:
        scaled = (action + 1) / self.num_actions
        return scaled * np.ones(self.old_obs_shape if self.is_obs_image else (1,)).astype(np.float32)

    def observation(self) -> np.ndarray:
        obs = np.stack(self._obs, axis=0).astype(np.float32)
        act = np.stack(self._act, axis=0).astype(np.float32)
        if self.is_obs_image:
            return np.concatenate([obs.reshape((-1,) + obs.shape[2:]), act], axis=0)
        return np.concatenate([obs, act], axis=1)

    def reset(self, **kwargs) -> np.ndarray:
        first = self.env.reset(**kwargs)
        self._obs = [first] * self.stack_history
        self._act = [self._plane(0)] * self.stack_history
        return self.observation()

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self._obs = [obs] + self._obs[:-1]
        self._act = [self._plane(action)] + self._act[:-1]
        return self.observation(), reward, done, info


class PlayerIdAndActionMaskWrapper:
    """`gym_env.py:356-365`: single-player environments present player ids 1 / 1 and an all-legal action mask."""

    def __init__(self, env) -> None:
        self.env = env
        self.current_player = self.opponent_player = 1
        self.actions_mask = np.ones(env.num_actions, dtype=np.bool_).flatten()

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, action):
        return self.env.step(action)


class _CartPolePhysics:
    """gym 0.23.1 CartPole-v1 (an un-vendored dependency of the reference; its published equations): float64 state, float32
    observation, reward 1 per step, failure beyond |x| 2.4 or |theta| 12 degrees, TimeLimit 500."""

    num_actions = 2
    observation_shape = (4,)

    def __init__(self, seed: int = 1) -> None:
        self._rs = np.random.RandomState(seed)
        self.state, self.steps, self.done = np.zeros(4), 0, True

    def reset(self, state=None) -> np.ndarray:
        self.state = np.asarray(state, np.float64).copy() if state is not None else self._rs.uniform(-0.05, 0.05, size=4)
        self.steps, self.done = 0, False
        return self.state.astype(np.float32)

    def step(self, action: int):
        if self.done:
            raise RuntimeError('Episode is over, call reset before using step method.')
        if action not in (0, 1):
            raise ValueError(f'Invalid action {action}')
        gravity, masscart, masspole, length, force_mag, tau = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
        total_mass, polemass_length = masspole + masscart, masspole * length
        x, x_dot, theta, theta_dot = self.state
        force = force_mag if action == 1 else -force_mag
        costheta, sintheta = np.cos(theta), np.sin(theta)
        temp = (force + polemass_length * theta_dot * theta_dot * sintheta) / total_mass
        thetaacc = (gravity * sintheta - costheta * temp) / (length * (4.0 / 3.0 - masspole * costheta * costheta / total_mass))
        xacc = temp - polemass_length * thetaacc * costheta / total_mass
        x, x_dot = x + tau * x_dot, x_dot + tau * xacc
        theta, theta_dot = theta + tau * theta_dot, theta_dot + tau * thetaacc
        self.state = np.array([x, x_dot, theta, theta_dot], np.float64)
        self.steps += 1
        failed = x < -2.4 or x > 2.4 or theta < -12 * 2 * np.pi / 360 or theta > 12 * 2 * np.pi / 360
        self.done = bool(failed or self.steps >= 500)
        return self.state.astype(np.float32), 1.0, self.done, {}


class CartPoleEnv(PlayerIdAndActionMaskWrapper):
    """Host-side CartPole-v1 as the reference's `create_classic_environment('CartPole-v1', stack_history=4)` presents it
    (`gym_env.py:436-459`): the physics above inside StackFrameAndAction(stack_history, is_obs_image=False) inside
    PlayerIdAndActionMaskWrapper.  Same rules as the device environment in `csrc/mz_env.h`; for evaluators and tools."""

    def __init__(self, stack_history: int = 4, seed: int = 1) -> None:
        super().__init__(StackFrameAndAction(_CartPolePhysics(seed), stack_history, False))
        self.stack_history = stack_history
        self.reset()

    def reset(self, state=None) -> np.ndarray:
        return self.env.reset(state=state)
