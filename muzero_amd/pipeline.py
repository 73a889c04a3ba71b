"""Host-side mirror of the self-play half of muzero/pipeline.py.

The reference runs one Python actor process per environment (pipeline.py:41-167): per step a `uct_search`, an
`env.step`, a trajectory append, and at episode end target computation + `make_unroll_sequence` + `data_queue.put`.
Here the per-step part (search, action sampling, env.step, record, auto-reset) is one lock-step move of thousands of
device-resident environments inside the HIP planner (`mz_selfplay_step`); this module only drains the device record
ring, cuts it into episodes and emits the same `(Transition, priority)` stream.

Functions with a reference counterpart keep its name, arguments and error behaviour:
compute_n_step_target (pipeline.py:632-673), compute_mc_return_target (:676-707), make_unroll_sequence (:710-767),
create_checkpoint / load_checkpoint (:802-807), run_self_play (:41-167).
"""
import collections
import copy
import os
import time
from typing import Any, Iterable, List, Mapping, NamedTuple, Optional, Text

import numpy as np


class Transition(NamedTuple):
    """Same fields, same order as replay.py:27-32 (the wire format between actors and the learner)."""
    state: Optional[np.ndarray]
    action: Optional[np.ndarray]
    pi_prob: Optional[np.ndarray]
    value: Optional[np.ndarray]
    reward: Optional[np.ndarray]


def compute_n_step_target(rewards: List[float], root_values: List[float], td_steps: int, discount: float) -> List[float]:
    """z_t = sum_{i<n} discount^i r_{t+i} + discount^n v_{t+n}, zero padded past the end (pipeline.py:632-673).
    Sums run left to right in float64 like the reference's Python `sum`, so results are bit-identical."""
    if len(rewards) != len(root_values):
        raise ValueError('Arguments `rewards` and `root_values` don have the same length.')
    T = len(rewards)
    r = list(rewards) + [0] * td_steps
    v = list(root_values) + [0] * td_steps
    pw = [discount**i for i in range(td_steps + 1)]
    out = []
    for t in range(T):
        acc = 0
        for i in range(td_steps):
            acc = acc + pw[i] * r[t + i]
        out.append(acc + pw[td_steps] * v[t + td_steps])
    return out


def compute_mc_return_target(rewards: List[float], player_ids: List[float]) -> List[float]:
    """Board games: +/- final reward from each mover's perspective, all zeros for a draw (pipeline.py:676-707)."""
    if len(rewards) != len(player_ids):
        raise ValueError('Arguments `rewards` and `player_ids` don have the same length.')
    T = len(rewards)
    out = [0.0] * T
    final_reward, final_player = rewards[-1], player_ids[-1]
    if final_reward != 0.0:
        for t in range(T):
            out[t] = final_reward if player_ids[t] == final_player else -final_reward
    return out


def make_unroll_sequence(observations, actions, rewards, pi_probs, values, priorities, unroll_steps) -> Iterable:
    """Yield (Transition, priority) per step with K-step stacked action / reward / value / policy; steps past the end are
    absorbing (action 0, reward 0, value 0, uniform policy) (pipeline.py:710-767).

    Differences from the reference, both deliberate: the caller's lists are NOT mutated (the reference extends them in
    place, pipeline.py:739-747), and actions are stored as int8 only when they fit (A <= 128) -- the reference's
    unconditional int8 cannot represent Gomoku 15x15 actions; larger action spaces get int16."""
    T = len(observations)
    K = unroll_steps
    n_act = len(pi_probs[-1])
    acts = list(actions) + ([0] * K if len(actions) == T else [])
    rews = list(rewards) + ([0] * K if len(rewards) == T else [])
    vals = list(values) + ([0] * K if len(values) == T else [])
    pis = list(pi_probs) + ([np.ones_like(pi_probs[-1]) / n_act] * K if len(pi_probs) == T else [])
    assert len(acts) == len(rews) == len(vals) == len(pis) == T + K
    act_dtype = np.int8 if n_act <= 128 else np.int16
    for t in range(T):
        yield (
            Transition(
                state=observations[t],
                action=np.array(acts[t:t + K], dtype=act_dtype),
                reward=np.array(rews[t:t + K], dtype=np.float32),
                value=np.array(vals[t:t + K], dtype=np.float32),
                pi_prob=np.array(pis[t:t + K], dtype=np.float32),
            ),
            priorities[t],
        )


def _compact(obj):
    """Tensors that are views of a larger storage are cloned: torch.save writes a tensor's WHOLE storage, and the parameters of a module that
    hip_learner.HipLearner adopted are views of one flat vector that also holds Adam's moments and the gradient slices -- a checkpoint of such
    a module would otherwise carry that vector (4x-33x the weights) inside its 'network' entry."""
    import torch

    if torch.is_tensor(obj):
        return obj.detach().clone() if obj.untyped_storage().nbytes() > obj.numel() * obj.element_size() else obj
    # only plain containers are rebuilt; everything else (namedtuples, defaultdicts, user classes) passes through unchanged.  An OrderedDict --
    # network.state_dict() -- is shallow-copied and its values replaced in place, so its `_metadata` attribute (the per-module version
    # info torch.save writes and load_state_dict reads) stays with it
    if isinstance(obj, dict) and type(obj) in (dict, collections.OrderedDict):
        out = copy.copy(obj)
        for k, v in obj.items():
            out[k] = _compact(v)
        if hasattr(obj, '_metadata'):
            out._metadata = obj._metadata
        return out
    if type(obj) in (list, tuple):
        return type(obj)(_compact(v) for v in obj)
    return obj


def create_checkpoint(state_to_save: Mapping[Text, Any], ckpt_file: str) -> None:
    """pipeline.py:802-803: torch.save of {'network', 'optimizer', 'lr_scheduler', 'train_steps'}."""
    import torch

    torch.save(_compact(state_to_save), ckpt_file)


def load_checkpoint(ckpt_file: str, device) -> Mapping[Text, Any]:
    """pipeline.py:806-807 (weights_only=False: the reference checkpoints hold optimizer / scheduler state dicts)."""
    import torch

    return torch.load(ckpt_file, map_location=torch.device(device), weights_only=False)


# ------------------------------------------------------------------------------------------------------------------
# multi-GPU sharding: environments are independent, so ranks own disjoint env ranges and never exchange data
# ------------------------------------------------------------------------------------------------------------------
def shard_range(total_envs: int, rank: int, world_size: int):
    """Contiguous env id range [lo, hi) of `rank` (sizes differ by at most one)."""
    if not 0 <= rank < world_size:
        raise ValueError(f'rank {rank} outside world of size {world_size}')
    base, extra = divmod(total_envs, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def aggregate_throughput(local_units: float, local_seconds: float):
    """Whole-job rate: total units of all ranks / slowest rank's time (the bench.py contract).  Uses the default
    torch.distributed group when initialised (RCCL on GPUs, gloo in the CPU tests); a single process otherwise."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return local_units / local_seconds, local_units, local_seconds
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    units = torch.tensor([float(local_units)], dtype=torch.float64, device=dev)
    secs = torch.tensor([float(local_seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(units, op=dist.ReduceOp.SUM)
    dist.all_reduce(secs, op=dist.ReduceOp.MAX)
    return float(units.item() / secs.item()), float(units.item()), float(secs.item())


def weights_key(network, train_steps_counter=None, config=None):
    """What an actor watches to know that the learner refreshed `network` (pipeline.py:261-267: every checkpoint_interval
    train steps).  In one process the tensors' version counters change; in another process (the reference's layout:
    shared-memory parameters, mp.Process actors) only the storage changes, so the signal that travels is the network's
    `weights_epoch` -- a shared-memory buffer the learner bumps AFTER `load_state_dict` (`MuZeroNet.publish_weights`).
    (Round 2 keyed on the train-step counter crossing a checkpoint_interval boundary: that fires BEFORE the learner has
    written the checkpoint and copied the weights, so a remote actor reloaded the old values and stayed one checkpoint
    behind.)  `train_steps_counter` / `config` are accepted for the old call sites and ignored."""
    epoch = getattr(network, 'weights_epoch', None)
    return network._weights_version(), (int(epoch.item()) if epoch is not None else 0)


# ------------------------------------------------------------------------------------------------------------------
# self-play actor over the device-resident planner
# ------------------------------------------------------------------------------------------------------------------
class EpisodeAssembler:
    """Cuts the lock-step record stream [moves, envs, ...] into per-env episodes and turns finished episodes into
    (Transition, priority) items exactly like pipeline.py:144-165."""

    def __init__(self, config, num_envs: int, obs_shape=None):
        self.config = config
        self.open = [[] for _ in range(num_envs)]
        self.obs_shape = None if obs_shape is None else tuple(obs_shape)  # the env's observation shape (records are flat rows)

    def feed(self, rec) -> Iterable:
        cfg = self.config
        n_moves, B = rec['action'].shape
        for m in range(n_moves):
            for b in range(B):
                o = rec['obs'][m, b] if self.obs_shape is None else rec['obs'][m, b].reshape(self.obs_shape)
                self.open[b].append((o, int(rec['action'][m, b]), float(rec['reward'][m, b]), rec['pi'][m, b],
                                     float(rec['root_value'][m, b]), int(rec['player'][m, b])))
                # same order as the reference's loop body: the mid-episode flush check first (pipeline.py:118-142), then the
                # end of the episode (:144-165) -- when both fall on one step the prefix is flushed and the rest finishes
                if (not cfg.is_board_game) and len(self.open[b]) == cfg.acc_seq_length + cfg.unroll_steps + cfg.td_steps:
                    yield from self._flush_prefix(b)
                if rec['done'][m, b]:
                    traj, self.open[b] = self.open[b], []
                    yield from self._finish(traj)

    def _finish(self, traj):
        cfg = self.config
        obs, actions, rewards, pis, roots, players = map(list, zip(*traj))
        if cfg.is_board_game:
            targets = compute_mc_return_target(rewards, players)
        else:
            targets = compute_n_step_target(rewards, roots, cfg.td_steps, cfg.discount)
        prios = np.abs(np.array(roots) - np.array(targets))
        yield from make_unroll_sequence(obs, actions, rewards, pis, targets, prios, cfg.unroll_steps)

    def _flush_prefix(self, b):
        cfg = self.config
        n = cfg.acc_seq_length
        obs, actions, rewards, pis, roots, _ = map(list, zip(*self.open[b]))
        targets = compute_n_step_target(rewards, roots, cfg.td_steps, cfg.discount)
        prios = np.abs(np.array(roots) - np.array(targets))
        k = n + cfg.unroll_steps
        yield from make_unroll_sequence(obs[:n], actions[:k], rewards[:k], pis[:k], targets[:k], prios[:k], cfg.unroll_steps)
        del self.open[b][:n]


def run_self_play(config, rank, network, device, env, data_queue, train_steps_counter, stop_event, tag: str = None,
                  moves_per_drain: int = 16, max_moves: Optional[int] = None) -> int:
    """Self-play until `stop_event` is set (pipeline.py:41-167).  `env` names a device environment ('CartPole-v1',
    'TicTacToe', 'Gomoku', or 'Synthetic-Atari': random frames standing in for the absent emulator); `config.num_envs` of
    them advance in lock-step on GPU `device`.  `data_queue` is either a queue -- items put on it are the reference's
    `(Transition, priority)` tuples, assembled on the host from the device records -- or a
    `muzero_amd.replay.PrioritizedReplay(device='cuda')`: then the planner's DEVICE EPILOGUE builds the items on the GPU and
    writes them straight into that replay (no host assembly, no queue, no data collector thread); only rewards and done
    flags are read back for the episode statistics.  Returns the number of env steps played."""
    from muzero_amd import planner as pl

    kinds = {'CartPole-v1': pl.ENV_CARTPOLE, 'TicTacToe': pl.ENV_TICTACTOE, 'Gomoku': pl.ENV_GOMOKU, 'Synthetic-Atari': pl.ENV_SYNTHETIC}
    name = env if isinstance(env, str) else getattr(env, 'name', None) or getattr(getattr(env, 'spec', None), 'id', None)
    if name not in kinds:
        raise ValueError(f'no device environment for {name!r}; available: {sorted(kinds)}')
    num_envs = int(getattr(config, 'num_envs', 1))
    idx = device.index if getattr(device, 'index', None) is not None else 0
    p = pl.Planner(pl.make_mz_config(network.planner_spec(), config, num_envs=num_envs, seed=int(getattr(config, 'planner_seed', 1)) + 7919 * rank), idx)
    p.load_state_dict(network.state_dict())
    from muzero_amd.replay import PrioritizedReplay

    on_device = isinstance(data_queue, PrioritizedReplay)
    if on_device:  # ONE device writer per replay (attach_device_writer raises on a second one): every actor rank its own shard
        p.attach_replay(data_queue, config, obs_shape=getattr(network, 'input_shape', None))
    p.selfplay_reset(kinds[name])
    asm = EpisodeAssembler(config, num_envs, getattr(network, 'input_shape', None))
    from muzero_amd import metrics as mzm

    tracker = mzm.ActorMetrics(mzm.run_file(config, f'actor{rank}', tag), num_envs)  # trackers.py:74-80 tag names

    version = weights_key(network)
    played = 0
    while not stop_event.is_set() and (max_moves is None or played < max_moves):
        key = weights_key(network)  # read BEFORE copying: a publish that lands during the copy makes the next check fire again
        if key != version:  # learner pushed new weights (pipeline.py:266)
            p.load_state_dict(network.state_dict())
            version = key
        n = moves_per_drain if max_moves is None else min(moves_per_drain, max_moves - played)
        # classic/atari schedules depend on train steps only; board games on the env's own step count (config.py:236-267)
        T = -1.0 if config.is_board_game else float(config.visit_softmax_temperature_fn(0, train_steps_counter.value))
        p.selfplay_step(T, n)
        if on_device:
            rec = p.selfplay_read(n, fields=('reward', 'done'))
            tracker.moves(rec['reward'], rec['done'])
        else:
            rec = p.selfplay_read(n)
            tracker.moves(rec['reward'], rec['done'])
            for item in asm.feed(rec):
                data_queue.put(item)
        played += n
    tracker.close()
    p.close()  # (detaches the device epilogue: the replay's write cursor and priorities go back to its host side)
    return played * num_envs


def run_board_game_evaluator(config, old_checkpoint_network, new_ckpt_network, device, env, temperature, checkpoint_files, stop_event,
                             initial_elo: int = -2000, tag: str = None, on_result=None) -> float:
    """pipeline.py:289-397: for every new checkpoint, one deterministic game new (black) vs previous (white) checkpoint on a
    host `games.BoardGameEnv`, searches through the HIP planner; Elo update as the reference does (white inherits black's
    rating).  `on_result(black_elo, env.steps, train_steps)` stands in for the tensorboard trackers.  Returns the final Elo."""
    from muzero_amd import mcts
    from muzero_amd.games import BoardGameEnv
    from muzero_amd.rating import compute_elo_rating

    if not isinstance(env, BoardGameEnv):
        raise ValueError(f'Expect env to be a valid BoardGameEnv instance, got {env}')
    for net in (old_checkpoint_network, new_ckpt_network):
        for p in net.parameters():
            p.requires_grad = False
    black_elo = white_elo = initial_elo
    from muzero_amd import metrics as mzm

    tracker = mzm.EvaluatorMetrics(mzm.run_file(config, 'evaluator', tag))  # trackers.py:173-189 tag names
    while True:
        if stop_event.is_set() and len(checkpoint_files) == 0:
            break
        if len(checkpoint_files) == 0:
            time.sleep(0.001)
            continue
        loaded_state = load_checkpoint(checkpoint_files.pop(0), device)
        new_ckpt_network.load_state_dict(loaded_state['network'])
        train_steps = loaded_state['train_steps']
        new_ckpt_network.eval()
        old_checkpoint_network.eval()
        obs = env.reset()
        done = False
        while not done:
            network = new_ckpt_network if env.current_player == env.black_player_id else old_checkpoint_network
            action, *_ = mcts.uct_search(state=obs, network=network, device=device, config=config, temperature=temperature,
                                         actions_mask=env.actions_mask, current_player=env.current_player, opponent_player=env.opponent_player,
                                         deterministic=True)
            obs, _, done, _ = env.step(action)
        if env.winner == env.black_player_id:
            black_elo, _ = compute_elo_rating(0, black_elo, white_elo)
        elif env.winner == env.white_player_id:
            black_elo, _ = compute_elo_rating(1, black_elo, white_elo)
        white_elo = black_elo
        tracker.board_game_step(black_elo, env.steps, train_steps)
        if on_result is not None:
            on_result(black_elo, env.steps, train_steps)
        old_checkpoint_network.load_state_dict(new_ckpt_network.state_dict())
    tracker.close()
    return black_elo


def run_evaluator(config, new_ckpt_network, device, env, temperature, checkpoint_files, stop_event, tag: str = None, num_episodes: int = 1,
                  on_result=None) -> List:
    """pipeline.py:400-488: for every new checkpoint, `num_episodes` deterministic episodes on a host environment exposing
    `reset / step / actions_mask / current_player / opponent_player` (e.g. `games.CartPoleEnv`), searches through the HIP
    planner.  `on_result(eval_returns, eval_steps, train_steps)` stands in for the tensorboard trackers; the list of those
    triples is returned."""
    from muzero_amd import mcts

    for p in new_ckpt_network.parameters():
        p.requires_grad = False
    from muzero_amd import metrics as mzm

    tracker = mzm.EvaluatorMetrics(mzm.run_file(config, 'evaluator', tag))  # trackers.py:165-170 tag names
    results = []
    while True:
        if stop_event.is_set() and len(checkpoint_files) == 0:
            break
        if len(checkpoint_files) == 0:
            time.sleep(0.001)
            continue
        loaded_state = load_checkpoint(checkpoint_files.pop(0), device)
        new_ckpt_network.load_state_dict(loaded_state['network'])
        train_steps = loaded_state['train_steps']
        new_ckpt_network.eval()
        eval_returns, eval_steps = [], []
        for _ in range(num_episodes):
            obs = env.reset()
            done, steps, returns = False, 0, 0.0
            while not done:
                action, *_ = mcts.uct_search(state=obs, network=new_ckpt_network, device=device, config=config, temperature=temperature,
                                             actions_mask=env.actions_mask, current_player=env.current_player,
                                             opponent_player=env.opponent_player, deterministic=True)
                obs, reward, done, _ = env.step(action)
                steps += 1
                returns += reward
            eval_returns.append(returns)
            eval_steps.append(steps)
        results.append((eval_returns, eval_steps, train_steps))
        tracker.step(eval_returns, eval_steps, train_steps)
        if on_result is not None:
            on_result(eval_returns, eval_steps, train_steps)
    tracker.close()
    return results


def rank_env() -> tuple:
    """(rank, local_rank, world_size) from the torchrun environment."""
    return int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
