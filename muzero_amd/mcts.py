"""Host-side mirror of muzero/mcts.py: same function names, arguments and error behaviour; the search itself runs in
the HIP planner (libmzplanner_hip.so).  `uct_search` is the drop-in (one root); `batched_uct_search` searches many
roots in lock-step on the GPU, which is the shape the planner is built for.

Randomness: the reference draws from the process-global numpy RNG inside the search (mcts.py:124,245,404).  Here the
draws are explicit inputs: by default the planner generates them on device (Philox keyed by planner seed, env id and
move counter); tests inject recorded draws through `rng=dict(noise=..., u_tie=..., u_final=...)`.
"""
from typing import Optional, Tuple

import numpy as np

from muzero_amd.config import KnownBounds, MuZeroConfig  # noqa: F401  (re-exported like mcts.py:27)

MAXIMUM_FLOAT_VALUE = float('inf')


class MinMaxStats:
    """mcts.py:33-48 (host mirror; the planner keeps one (min, max) pair per env in LDS)."""

    def __init__(self, known_bounds: Optional[KnownBounds]):
        self.maximum = known_bounds.max if known_bounds else -MAXIMUM_FLOAT_VALUE
        self.minimum = known_bounds.min if known_bounds else MAXIMUM_FLOAT_VALUE

    def update(self, value: float):
        self.maximum = max(self.maximum, value)
        self.minimum = min(self.minimum, value)

    def normalize(self, value: float) -> float:
        if self.maximum > self.minimum:
            return (value - self.minimum) / (self.maximum - self.minimum)
        return value


def add_dirichlet_noise(prob: np.ndarray, eps: float = 0.25, alpha: float = 0.03, noise: Optional[np.ndarray] = None):
    """mcts.py:220-247.  `noise` (a Dirichlet sample) may be injected; by default it is drawn from np.random like the
    reference.  Same validation and dtype behaviour: float32 prior in, float64 noised prior out."""
    if not isinstance(prob, np.ndarray) or prob.dtype not in (np.float32, np.float64):
        raise ValueError(f"Expect `prob` to be a numpy.array, got {prob}")
    if not isinstance(eps, float) or not 0.0 <= eps <= 1.0:
        raise ValueError(f"Expect `eps` to be a float in the range [0.0, 1.0], got {eps}")
    if not isinstance(alpha, float) or not 0.0 <= alpha <= 1.0:
        raise ValueError(f"Expect `alpha` to be a float in the range [0.0, 1.0], got {alpha}")
    if noise is None:
        noise = np.random.dirichlet(np.ones_like(prob) * alpha)
    return (1 - eps) * prob + eps * noise


def generate_play_policy(visits_count: np.ndarray, temperature: float) -> np.ndarray:
    """mcts.py:250-280: visits ** clip(1/T, 1, 5), normalised; T == 0 is linear in the visit counts."""
    if not isinstance(visits_count, np.ndarray) or len(visits_count.shape) != 1 or visits_count.shape == (0,):
        raise ValueError(f"Expect `visits_count` to be a 1D numpy.array, got {visits_count}")
    if not isinstance(temperature, float) or not 0.0 <= temperature <= 1.0:
        raise ValueError(f"Expect `temperature` to be float type in the range [0.0, 1.0], got {temperature}")
    v = np.asarray(visits_count, dtype=np.int64)
    if temperature > 0.0:
        v = np.power(v, max(1.0, min(5.0, 1.0 / temperature)))
    return v / np.sum(v)


def set_illegal_action_probs_to_zero(actions_mask: np.ndarray, prob: np.ndarray) -> np.ndarray:
    """mcts.py:283-299"""
    assert actions_mask.shape == prob.shape
    prob = np.where(actions_mask, prob, 0.0)
    total = np.sum(prob)
    if total > 0:
        prob /= total
    return prob


def _planner_for(network, config, num_envs, device):
    """One search planner per (network, search config, capacity), cached on the network object and re-fed with weights
    whenever the module's parameters changed."""
    from muzero_amd import planner as _pl

    kb = config.known_bounds
    key = (config.num_simulations, config.discount, config.pb_c_base, config.pb_c_init, bool(config.is_board_game),
           None if kb is None else (kb.min, kb.max), config.root_dirichlet_alpha, config.root_exploration_eps, int(num_envs))
    cache = network.__dict__.setdefault('_search_planners', {})
    entry = cache.get(key)
    version = network._weights_version()
    if entry is None:
        idx = device.index if getattr(device, 'index', None) is not None else 0
        if getattr(device, 'type', 'cuda') != 'cuda':
            raise _pl.PlannerError(f'uct_search runs on the HIP planner only; got device {device} (no CPU fallback)')
        pl = _pl.Planner(_pl.make_mz_config(network.planner_spec(), config, num_envs=num_envs, seed=getattr(config, 'planner_seed', 1)), idx)
        entry = [pl, None]
        cache[key] = entry
    if entry[1] != version:
        entry[0].load_state_dict(network.state_dict())
        entry[1] = version
    return entry[0]


def batched_uct_search(states, network, device, config, temperature, actions_mask, current_player, opponent_player, deterministic=False,
                       rng=None):
    """Lock-step search of B roots.  states [B, *obs]; actions_mask [B, A] bool or None; players/temperature scalars or
    [B].  Returns (actions int32 [B], pi_probs float64 [B, A], root_values float64 [B])."""
    if config.is_board_game:
        assert config.discount == 1.0
    states = np.asarray(states)
    b = states.shape[0]
    pl = _planner_for(network, config, max(b, getattr(config, 'num_envs', 1)), device)
    rng = rng or {}
    out = pl.search(states.reshape(b, -1), actions_mask, current_player, opponent_player, temperature, deterministic,
                    noise=rng.get('noise'), u_tie=rng.get('u_tie'), u_final=rng.get('u_final'))
    return out['action'], out['pi'], out['root_value']


def uct_search(state, network, device, config, temperature, actions_mask, current_player, opponent_player, deterministic=False,
               rng=None) -> Tuple[int, np.ndarray, float]:
    """Drop-in for mcts.py:302-407: one root in, (action, pi_prob, root_value) out."""
    if not isinstance(temperature, float) or not 0.0 <= temperature <= 1.0:
        raise ValueError(f"Expect `temperature` to be float type in the range [0.0, 1.0], got {temperature}")
    mask = None if actions_mask is None else np.asarray(actions_mask)[None, ...]
    a, pi, v = batched_uct_search(np.asarray(state)[None, ...], network, device, config, temperature, mask, current_player, opponent_player,
                                  deterministic, rng)
    if not deterministic and np.isnan(pi[0]).any():
        # every visit fell on illegal root children: the reference's 0/0 policy makes np.random.choice raise (mcts.py:279,404)
        raise ValueError('probabilities contain NaN')
    return int(a[0]), pi[0], float(v[0])
