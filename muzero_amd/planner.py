"""ctypes binding of libmzplanner_hip.so (C ABI: include/mzplanner.h).

This module is the only place where the Python mirror of the reference API touches native code.  There is no CPU
fallback: if the library cannot be loaded, or no MI355X is visible, the calls raise PlannerError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libmzplanner_hip.so')

NET_MLP, NET_BOARD, NET_ATARI = 0, 1, 2
ENV_NONE, ENV_CARTPOLE, ENV_TICTACTOE, ENV_GOMOKU, ENV_SYNTHETIC = 0, 1, 2, 3, 4
_NET_KINDS = {'mlp': NET_MLP, 'board': NET_BOARD, 'atari': NET_ATARI}

# every symbol include/mzplanner.h declares (tests/test_abi.py checks the library exports all of them)
ABI_SYMBOLS = [
    'mz_last_error', 'mz_version', 'mz_planner_describe', 'mz_planner_create', 'mz_planner_destroy', 'mz_planner_set_param', 'mz_planner_commit_params',
    'mz_planner_initial_inference', 'mz_planner_recurrent_inference', 'mz_planner_hidden_size', 'mz_planner_search',
    'mz_planner_search_scripted', 'mz_selfplay_reset', 'mz_selfplay_step', 'mz_selfplay_read', 'mz_selfplay_counters',
    'mz_selfplay_attach_replay',
    'mz_profile_begin', 'mz_profile_end', 'mz_planner_synchronize',
]


class PlannerError(RuntimeError):
    pass


class MzConfig(C.Structure):
    _fields_ = [
        ('net_kind', C.c_int32), ('obs_c', C.c_int32), ('obs_h', C.c_int32), ('obs_w', C.c_int32), ('num_actions', C.c_int32),
        ('num_planes', C.c_int32), ('hidden_dim', C.c_int32), ('num_res_blocks', C.c_int32), ('value_support_size', C.c_int32),
        ('reward_support_size', C.c_int32), ('num_simulations', C.c_int32), ('discount', C.c_double), ('pb_c_base', C.c_double),
        ('pb_c_init', C.c_double), ('is_board_game', C.c_int32), ('has_known_bounds', C.c_int32), ('known_bounds_min', C.c_double),
        ('known_bounds_max', C.c_double), ('root_dirichlet_alpha', C.c_double), ('root_exploration_eps', C.c_double),
        ('num_envs', C.c_int32), ('max_ties', C.c_int32), ('seed', C.c_uint64), ('legacy_scalar_promotion', C.c_int32),
    ]


class MzRngInputs(C.Structure):
    _fields_ = [('h_noise', C.c_void_p), ('h_u_tie', C.c_void_p), ('h_u_final', C.c_void_p)]


class MzReplayRing(C.Structure):
    _fields_ = [('capacity', C.c_int64), ('state', C.c_void_p), ('action', C.c_void_p), ('pi_prob', C.c_void_p), ('value', C.c_void_p),
                ('reward', C.c_void_p), ('priority', C.c_void_p), ('num_added', C.c_void_p), ('origin', C.c_void_p),
                ('acc_seq_length', C.c_int32), ('unroll_steps', C.c_int32), ('td_steps', C.c_int32)]


_lib = None


def load_library():
    """dlopen the planner library and declare its prototypes.  Raises PlannerError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PlannerError(
            f'{LIB_PATH} not found: build it with `python -m muzero_amd.build` (hipcc, gfx950). The planning path has no CPU fallback.'
        )
    import torch  # noqa: F401  -- torch bundles its own HIP runtime: load it first so ONE libamdhip64 serves the process
    L = C.CDLL(LIB_PATH)
    vp, i32, i64p = C.c_void_p, C.c_int32, C.POINTER(C.c_int64)
    L.mz_last_error.restype = C.c_char_p
    L.mz_version.restype = C.c_char_p
    L.mz_planner_describe.restype = C.c_char_p
    L.mz_planner_describe.argtypes = [C.c_void_p]
    L.mz_planner_create.argtypes = [C.POINTER(MzConfig), C.c_int, C.POINTER(vp)]
    L.mz_planner_destroy.argtypes = [vp]
    L.mz_planner_set_param.argtypes = [vp, C.c_char_p, vp, i64p, i32]
    L.mz_planner_commit_params.argtypes = [vp]
    L.mz_planner_initial_inference.argtypes = [vp, i32, vp, vp, vp, vp]
    L.mz_planner_recurrent_inference.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    L.mz_planner_hidden_size.argtypes = [vp]
    L.mz_planner_hidden_size.restype = i32
    L.mz_planner_search.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, C.POINTER(MzRngInputs), vp, vp, vp, vp]
    L.mz_planner_search_scripted.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, C.POINTER(MzRngInputs), vp, vp, vp, vp, vp, vp]
    L.mz_selfplay_reset.argtypes = [vp, i32, vp]
    L.mz_selfplay_step.argtypes = [vp, C.c_double, i32]
    L.mz_selfplay_read.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.mz_selfplay_counters.argtypes = [vp, i64p]
    L.mz_selfplay_attach_replay.argtypes = [vp, C.POINTER(MzReplayRing)]
    L.mz_profile_begin.argtypes = [vp]
    L.mz_profile_end.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.mz_planner_synchronize.argtypes = [vp]
    for name in ABI_SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int:
            fn.restype = C.c_int
    _lib = L
    return L


def _chk(rc):
    if rc != 0:
        raise PlannerError(f'mzplanner error {rc}: {load_library().mz_last_error().decode()}')


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_mz_config(spec, config=None, num_envs=1, max_ties=0, seed=1, **search_overrides):
    """Build an mz_config from a network spec (MuZeroNet.planner_spec()) and a MuZeroConfig-like object."""
    shape = tuple(spec['input_shape'])
    if spec['kind'] == 'mlp':
        c, h, w = int(np.prod(shape)), 1, 1
    else:
        c, h, w = shape
    g = lambda name, default: search_overrides.get(name, getattr(config, name, default) if config is not None else default)  # noqa: E731
    kb = g('known_bounds', None)
    return MzConfig(
        net_kind=_NET_KINDS[spec['kind']], obs_c=c, obs_h=h, obs_w=w, num_actions=spec['num_actions'], num_planes=spec['num_planes'],
        hidden_dim=spec['hidden_dim'], num_res_blocks=spec['num_res_blocks'], value_support_size=spec['value_support_size'],
        reward_support_size=spec['reward_support_size'], num_simulations=int(g('num_simulations', 1)), discount=float(g('discount', 1.0)),
        pb_c_base=float(g('pb_c_base', 19652)), pb_c_init=float(g('pb_c_init', 1.25)), is_board_game=int(bool(g('is_board_game', False))),
        has_known_bounds=int(kb is not None), known_bounds_min=float(kb[0]) if kb is not None else 0.0,
        known_bounds_max=float(kb[1]) if kb is not None else 0.0, root_dirichlet_alpha=float(g('root_dirichlet_alpha', 0.25)),
        root_exploration_eps=float(g('root_exploration_eps', 0.25)), num_envs=int(num_envs), max_ties=int(max_ties), seed=int(seed),
        legacy_scalar_promotion=int(bool(g('legacy_scalar_promotion', False))),
    )


class Planner:
    """One planner handle on one GPU (mz_planner*)."""

    def __init__(self, mz_config, device_id=0):
        self.lib = load_library()
        self.cfg = mz_config
        h = C.c_void_p()
        _chk(self.lib.mz_planner_create(C.byref(mz_config), int(device_id), C.byref(h)))
        self.h = h
        self.A = mz_config.num_actions
        self.S = mz_config.num_simulations
        self.B = mz_config.num_envs
        self.obs_dim = mz_config.obs_c * mz_config.obs_h * mz_config.obs_w
        self.hidden_size = self.lib.mz_planner_hidden_size(self.h)
        self.max_ties = mz_config.max_ties if mz_config.max_ties > 0 else 4 * self.S + 8

    def close(self):
        if getattr(self, 'h', None):
            if getattr(self, '_replay_keepalive', None) is not None:
                try:
                    self.detach_replay()  # hand write cursor and priorities back: the replay outlives this planner
                except Exception:
                    pass
            self.lib.mz_planner_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights ----
    def load_state_dict(self, state_dict):
        for name, t in state_dict.items():
            if name.endswith('num_batches_tracked'):
                continue
            a = np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, 'detach') else t, dtype=np.float32)
            shape = (C.c_int64 * max(a.ndim, 1))(*(a.shape if a.ndim else (1,)))
            _chk(self.lib.mz_planner_set_param(self.h, name.encode(), _p(a), shape, max(a.ndim, 1)))
        _chk(self.lib.mz_planner_commit_params(self.h))

    # ---- inference (network.py:62-111) ----
    def initial_inference(self, obs):
        obs = np.ascontiguousarray(obs, np.float32).reshape(-1, self.obs_dim)
        b = obs.shape[0]
        hidden = np.empty((b, self.hidden_size), np.float32)
        pi = np.empty((b, self.A), np.float32)
        value = np.empty(b, np.float32)
        _chk(self.lib.mz_planner_initial_inference(self.h, b, _p(obs), _p(hidden), _p(pi), _p(value)))
        return hidden, pi, value

    def recurrent_inference(self, hidden, action):
        hidden = np.ascontiguousarray(hidden, np.float32).reshape(-1, self.hidden_size)
        action = np.ascontiguousarray(action, np.int32).reshape(-1)
        b = hidden.shape[0]
        out = np.empty((b, self.hidden_size), np.float32)
        reward = np.empty(b, np.float32)
        pi = np.empty((b, self.A), np.float32)
        value = np.empty(b, np.float32)
        _chk(self.lib.mz_planner_recurrent_inference(self.h, b, _p(hidden), _p(action), _p(out), _p(reward), _p(pi), _p(value)))
        return out, reward, pi, value

    # ---- search (mcts.py:302-407) ----
    def _rng(self, b, noise, u_tie, u_final):
        if noise is None and u_tie is None and u_final is None:
            return None, ()
        u_tie_a = np.full((b, self.max_ties), 0.5, np.float64)
        if u_tie is not None:
            u = np.asarray(u_tie, np.float64).reshape(b, -1)
            n = min(u.shape[1], self.max_ties)
            u_tie_a[:, :n] = u[:, :n]
        u_final_a = np.ascontiguousarray(np.broadcast_to(0.5 if u_final is None else u_final, (b,)), np.float64)
        noise_a = None if noise is None else np.ascontiguousarray(noise, np.float64).reshape(b, self.A)
        rng = MzRngInputs(noise_a.ctypes.data if noise_a is not None else None, u_tie_a.ctypes.data, u_final_a.ctypes.data)
        return rng, (u_tie_a, u_final_a, noise_a)

    def search(self, obs, mask, current_player, opponent_player, temperature, deterministic=False, noise=None, u_tie=None, u_final=None):
        obs = np.ascontiguousarray(obs, np.float32).reshape(-1, self.obs_dim)
        b = obs.shape[0]
        mask_a = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(b, self.A)
        cur = np.ascontiguousarray(np.broadcast_to(current_player, (b,)), np.int32)
        opp = np.ascontiguousarray(np.broadcast_to(opponent_player, (b,)), np.int32)
        temp = np.ascontiguousarray(np.broadcast_to(temperature, (b,)), np.float64)
        rng, keep = self._rng(b, noise, u_tie, u_final)
        action = np.empty(b, np.int32)
        pi = np.empty((b, self.A), np.float64)
        root = np.empty(b, np.float64)
        visits = np.empty((b, self.A), np.int32)
        _chk(self.lib.mz_planner_search(
            self.h, b, _p(obs), _p(mask_a), _p(cur), _p(opp), _p(temp), int(bool(deterministic)), C.byref(rng) if rng is not None else None,
            _p(action), _p(pi), _p(root), _p(visits),
        ))
        del keep
        return dict(action=action, pi=pi, root_value=root, visits=visits)

    def search_scripted(self, pi0, values, rewards, mask, current_player, opponent_player, temperature, deterministic=False, noise=None,
                        u_tie=None, u_final=None):
        pi0 = np.ascontiguousarray(pi0, np.float32).reshape(-1, self.A)
        b = pi0.shape[0]
        values = np.ascontiguousarray(values, np.float32).reshape(b, self.S)
        rewards = np.ascontiguousarray(rewards, np.float32).reshape(b, self.S)
        mask_a = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(b, self.A)
        cur = np.ascontiguousarray(np.broadcast_to(current_player, (b,)), np.int32)
        opp = np.ascontiguousarray(np.broadcast_to(opponent_player, (b,)), np.int32)
        temp = np.ascontiguousarray(np.broadcast_to(temperature, (b,)), np.float64)
        rng, keep = self._rng(b, noise, u_tie, u_final)
        action = np.empty(b, np.int32)
        pi = np.empty((b, self.A), np.float64)
        root = np.empty(b, np.float64)
        visits = np.empty((b, self.A), np.int32)
        tp = np.empty((b, self.S), np.int32)
        ta = np.empty((b, self.S), np.int32)
        _chk(self.lib.mz_planner_search_scripted(
            self.h, b, _p(pi0), _p(values), _p(rewards), _p(mask_a), _p(cur), _p(opp), _p(temp), int(bool(deterministic)),
            C.byref(rng) if rng is not None else None, _p(action), _p(pi), _p(root), _p(visits), _p(tp), _p(ta),
        ))
        del keep
        return dict(action=action, pi=pi, root_value=root, visits=visits, trace_parent=tp, trace_action=ta)

    # ---- device-resident self-play (pipeline.py:83-113) ----
    def selfplay_reset(self, env_kind, init_state=None):
        init = None if init_state is None else np.ascontiguousarray(init_state, np.float64).reshape(self.B, 4)
        _chk(self.lib.mz_selfplay_reset(self.h, int(env_kind), _p(init)))

    def selfplay_step(self, temperature=1.0, n_moves=1):
        _chk(self.lib.mz_selfplay_step(self.h, float(temperature), int(n_moves)))

    def selfplay_read(self, n_moves, fields=None):
        """Records of the last `n_moves` moves, arrays [n_moves, B, ...].  `fields`: subset of ('obs', 'action', 'reward', 'pi',
        'root_value', 'player', 'done') to copy out (default all): with the device epilogue attached a host loop only needs
        rewards and done flags for its episode statistics."""
        B, A, D = self.B, self.A, self.obs_dim
        spec = dict(obs=((n_moves, B, D), np.float32), action=((n_moves, B), np.int32), reward=((n_moves, B), np.float32),
                    pi=((n_moves, B, A), np.float64), root_value=((n_moves, B), np.float64), player=((n_moves, B), np.int32),
                    done=((n_moves, B), np.uint8))
        want = set(spec) if fields is None else set(fields)
        out = {k: np.empty(shp, dt) for k, (shp, dt) in spec.items() if k in want}
        ptr = [(_p(out[k]) if k in out else None) for k in ('obs', 'action', 'reward', 'pi', 'root_value', 'player', 'done')]
        _chk(self.lib.mz_selfplay_read(self.h, n_moves, *ptr))
        return out

    def attach_replay(self, replay, config, obs_shape=None, with_origin=False):
        """Device epilogue (mz_selfplay_attach_replay): finished trajectories become (Transition, priority) items in `replay`
        -- a `muzero_amd.replay.PrioritizedReplay(device='cuda')` -- on the GPU, with the reference's target / unroll-window
        arithmetic (pipeline.py:118-165, 632-767); the host only reads `replay.num_added`.  `config` supplies acc_seq_length,
        unroll_steps, td_steps.  Call before `selfplay_reset`.  `with_origin`: also record which env produced each item
        (returned tensor; tests)."""
        import torch

        K, A = int(config.unroll_steps), self.A
        shp = tuple(obs_shape) if obs_shape is not None else (self.obs_dim,)
        adt = torch.int8 if A <= 128 else torch.int16  # (int16 where the reference's int8 field overflows: pipeline.py:753, Gomoku 15x15)
        if replay._ring is None:
            replay.allocate(dict(state=shp, action=(K,), pi_prob=(K, A), value=(K,), reward=(K,)), dict(action=adt))
        ring = replay._ring
        if ring['state'].device.type != 'cuda' or ring['state'].dtype != torch.float32 or ring['action'].dtype != adt:
            raise PlannerError(f'attach_replay needs a replay on the GPU with float32 states and {adt} actions ({A} actions)')
        if int(np.prod(ring['state'].shape[1:])) != self.obs_dim or tuple(ring['pi_prob'].shape[1:]) != (K, A):
            raise PlannerError('replay item shapes do not match the planner (observation size, unroll_steps, num_actions)')
        prio, count = replay.attach_device_writer()
        origin = torch.full((replay.capacity,), -1, dtype=torch.int32, device=ring['state'].device) if with_origin else None
        r = MzReplayRing(replay.capacity, ring['state'].data_ptr(), ring['action'].data_ptr(), ring['pi_prob'].data_ptr(), ring['value'].data_ptr(),
                         ring['reward'].data_ptr(), prio.data_ptr(), count.data_ptr(), origin.data_ptr() if with_origin else None,
                         int(config.acc_seq_length), K, int(config.td_steps))
        torch.cuda.synchronize()
        _chk(self.lib.mz_selfplay_attach_replay(self.h, C.byref(r)))
        self._replay_keepalive = (replay, prio, count, origin)
        return origin

    def detach_replay(self):
        """Drains the planner's stream, detaches the epilogue and hands counter / priorities back to the replay's host side."""
        _chk(self.lib.mz_selfplay_attach_replay(self.h, None))
        if self._replay_keepalive is not None:
            self._replay_keepalive[0].detach_device_writer()
        self._replay_keepalive = None

    def selfplay_counters(self):
        c = (C.c_int64 * 4)()
        _chk(self.lib.mz_selfplay_counters(self.h, c))
        return dict(env_steps=c[0], simulations=c[1], episodes=c[2], episode_steps=c[3])

    # ---- measurement ----
    def synchronize(self):
        _chk(self.lib.mz_planner_synchronize(self.h))

    def profile_begin(self):
        _chk(self.lib.mz_profile_begin(self.h))

    def describe(self) -> str:
        """The kernel build the last search launch dispatched to + the diagnostic switches as this handle read them (mz_planner_describe)."""
        return self.lib.mz_planner_describe(self.h).decode()

    def profile_end(self):
        ms, kms, n = C.c_double(), C.c_double(), C.c_int64()
        _chk(self.lib.mz_profile_end(self.h, C.byref(ms), C.byref(kms), C.byref(n)))
        return dict(elapsed_ms=ms.value, search_kernel_ms=kms.value, search_kernel_launches=n.value)


class InferenceEngine:
    """Planner used only for MuZeroNet.initial_inference / recurrent_inference (network.py:62-111)."""

    def __init__(self, spec, device=None, num_envs=16):
        idx = 0
        if device is not None and getattr(device, 'type', 'cuda') != 'cuda':
            raise PlannerError(f'initial/recurrent inference run on the HIP planner only; got device {device} (no CPU fallback)')
        if device is not None and getattr(device, 'index', None) is not None:
            idx = device.index
        self.planner = Planner(make_mz_config(spec, None, num_envs=num_envs), idx)

    def load_state_dict(self, sd):
        self.planner.load_state_dict(sd)

    def initial_inference(self, obs):
        return self.planner.initial_inference(obs)

    def recurrent_inference(self, hidden, action):
        return self.planner.recurrent_inference(hidden, action)
