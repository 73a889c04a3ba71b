"""ctypes front-end of the CPU oracle (oracle/libmzoracle.so) + numpy restatements of the
list-based pipeline helpers.  TEST INFRASTRUCTURE ONLY: importable from tests/, from
__graft_entry__.smoke() and from bench.py's cpu_baseline leg -- never from muzero_amd/.

Parity status: pinned against tests/golden/*.npz (recorded from the reference by gen_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libmzoracle.so')


def build(force=False):
    src = [os.path.join(HERE, f) for f in ('mz_oracle.c', 'mz_oracle.h', 'Makefile')]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.check_call(['make', '-C', HERE, '-s', 'libmzoracle.so'])
    return LIB_PATH


class SearchConfig(C.Structure):
    _fields_ = [
        ('num_actions', C.c_int32),
        ('num_simulations', C.c_int32),
        ('discount', C.c_double),
        ('pb_c_base', C.c_double),
        ('pb_c_init', C.c_double),
        ('is_board_game', C.c_int32),
        ('has_known_bounds', C.c_int32),
        ('kb_min', C.c_double),
        ('kb_max', C.c_double),
        ('dirichlet_alpha', C.c_double),
        ('exploration_eps', C.c_double),
        ('legacy_scalar_promotion', C.c_int32),
    ]


class RngInputs(C.Structure):
    _fields_ = [('noise', C.c_void_p), ('u_tie', C.c_void_p), ('n_tie', C.c_int32), ('u_final', C.c_double)]


class SearchResult(C.Structure):
    _fields_ = [('action', C.c_int32), ('root_value', C.c_double), ('n_tie_used', C.c_int32), ('status', C.c_int32)]


class CartPole(C.Structure):
    _fields_ = [('s', C.c_double * 4), ('steps', C.c_int32), ('done', C.c_int32)]


MAX_BOARD = 19
MAX_STACK = 8


class Board(C.Structure):
    _fields_ = [
        ('board_size', C.c_int32),
        ('stack', C.c_int32),
        ('num_to_win', C.c_int32),
        ('num_actions', C.c_int32),
        ('board', C.c_int8 * (MAX_BOARD * MAX_BOARD)),
        ('mask', C.c_uint8 * (MAX_BOARD * MAX_BOARD + 1)),
        ('planes', C.c_int8 * (2 * MAX_STACK * MAX_BOARD * MAX_BOARD)),
        ('current_player', C.c_int32),
        ('steps', C.c_int32),
        ('winner', C.c_int32),
        ('last_action', C.c_int32 * 2),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.mzo_net_create_mlp.restype = C.c_void_p
        L.mzo_net_create_mlp.argtypes = [C.c_int32] * 6 + [C.c_void_p]
        L.mzo_net_create_conv.restype = C.c_void_p
        L.mzo_net_create_conv.argtypes = [C.c_int32] * 9 + [C.c_void_p, C.c_int32]
        L.mzo_net_create_scripted.restype = C.c_void_p
        L.mzo_net_create_scripted.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.mzo_net_destroy.argtypes = [C.c_void_p]
        L.mzo_net_hidden_size.argtypes = [C.c_void_p]
        L.mzo_net_obs_size.argtypes = [C.c_void_p]
        L.mzo_initial_inference.argtypes = [C.c_void_p] * 5
        L.mzo_recurrent_inference.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mzo_uct_search.restype = SearchResult
        L.mzo_uct_search.argtypes = [
            C.POINTER(SearchConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int32,
            C.POINTER(RngInputs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        ]
        L.mzo_uct_search_batch.restype = C.c_int32
        L.mzo_uct_search_batch.argtypes = [
            C.POINTER(SearchConfig), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
            C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
        ]
        L.mzo_prepare_root_prior.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_double, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.mzo_generate_play_policy.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_void_p]
        L.mzo_sample_action.restype = C.c_int32
        L.mzo_sample_action.argtypes = [C.c_void_p, C.c_int32, C.c_double]
        L.mzo_signed_parabolic.restype = C.c_float
        L.mzo_signed_parabolic.argtypes = [C.c_float]
        L.mzo_logits_to_value.restype = C.c_float
        L.mzo_logits_to_value.argtypes = [C.c_void_p, C.c_int32]
        L.mzo_normalize_hidden.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.mzo_expf.restype = C.c_float
        L.mzo_expf.argtypes = [C.c_float]
        L.mzo_cartpole_reset.argtypes = [C.POINTER(CartPole), C.c_void_p]
        L.mzo_cartpole_step.restype = C.c_double
        L.mzo_cartpole_step.argtypes = [C.POINTER(CartPole), C.c_int32, C.c_void_p]
        L.mzo_stack_reset.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
        L.mzo_stack_push.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]
        L.mzo_board_reset.argtypes = [C.POINTER(Board), C.c_int32, C.c_int32, C.c_int32]
        L.mzo_board_step.restype = C.c_int32
        L.mzo_board_step.argtypes = [C.POINTER(Board), C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        L.mzo_board_observation.argtypes = [C.POINTER(Board), C.c_void_p]
        L.mzo_board_game_over.restype = C.c_int32
        L.mzo_board_game_over.argtypes = [C.POINTER(Board)]
        L.mzo_n_step_target.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p]
        L.mzo_mc_return_target.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def make_config(num_actions, num_simulations, discount, is_board_game=False, known_bounds=None, dirichlet_alpha=0.25,
                exploration_eps=0.25, pb_c_base=19652, pb_c_init=1.25, legacy_scalar_promotion=False):
    return SearchConfig(
        int(num_actions), int(num_simulations), float(discount), float(pb_c_base), float(pb_c_init), int(bool(is_board_game)),
        int(known_bounds is not None), float(known_bounds[0]) if known_bounds is not None else 0.0,
        float(known_bounds[1]) if known_bounds is not None else 0.0, float(dirichlet_alpha), float(exploration_eps), int(bool(legacy_scalar_promotion)),
    )


class Net:
    """Owner of an mzo_net handle."""

    def __init__(self, handle, num_actions, keep=()):
        if not handle:
            raise RuntimeError('oracle network creation failed (parameter count mismatch?)')
        self.h = handle
        self.A = num_actions
        self._keep = keep
        self.hidden_size = lib().mzo_net_hidden_size(self.h)
        self.obs_size = lib().mzo_net_obs_size(self.h)

    def __del__(self):
        try:
            if self.h:
                lib().mzo_net_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @staticmethod
    def _param_array(tensors):
        arrs = [_f32(t) for t in tensors]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        return arrs, ptrs

    @classmethod
    def mlp(cls, state_dict, input_dim, num_actions, num_planes, hidden_dim, value_support, reward_support):
        tensors = [np.asarray(v) for v in state_dict.values()]
        assert len(tensors) == 20
        arrs, ptrs = cls._param_array(tensors)
        h = lib().mzo_net_create_mlp(input_dim, num_actions, num_planes, hidden_dim, value_support, reward_support, ptrs)
        return cls(h, num_actions, keep=(arrs, ptrs))

    @classmethod
    def conv(cls, state_dict, kind, input_shape, num_actions, num_res_blocks, num_planes, value_support=1, reward_support=1):
        tensors = [np.asarray(v) for k, v in state_dict.items() if not k.endswith('num_batches_tracked')]
        arrs, ptrs = cls._param_array(tensors)
        c, hh, ww = input_shape
        h = lib().mzo_net_create_conv(
            {'board': 0, 'atari': 1}[kind], c, hh, ww, num_actions, num_res_blocks, num_planes, value_support, reward_support, ptrs, len(arrs)
        )
        return cls(h, num_actions, keep=(arrs, ptrs))

    @classmethod
    def from_module(cls, net, kind):
        """The oracle network holding the weights of a `muzero_amd.network` module (`kind`: 'mlp' or 'conv'); the shapes come from the
        module's own `planner_spec()`.  What the parity tests, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` legs build their
        checker from (round 6: it used to live in tests/test_oracle_nets.py, which made bench.py import the test tree)."""
        sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
        spec = net.planner_spec()
        if kind == 'mlp':
            return cls.mlp(sd, int(np.prod(spec['input_shape'])), spec['num_actions'], spec['num_planes'], spec['hidden_dim'],
                           spec['value_support_size'], spec['reward_support_size'])
        return cls.conv(sd, spec['kind'], spec['input_shape'], spec['num_actions'], spec['num_res_blocks'], spec['num_planes'],
                        spec['value_support_size'], spec['reward_support_size'])

    @classmethod
    def scripted(cls, pi0, values, rewards):
        pi0, values, rewards = _f32(pi0), _f32(values), _f32(rewards)
        h = lib().mzo_net_create_scripted(len(pi0), _p(pi0), _p(values), _p(rewards), len(values))
        return cls(h, len(pi0), keep=(pi0, values, rewards))

    def initial_inference(self, obs):
        obs = _f32(obs).reshape(-1)
        hidden = np.zeros(self.hidden_size, np.float32)
        pi = np.zeros(self.A, np.float32)
        v = np.zeros(1, np.float32)
        lib().mzo_initial_inference(self.h, _p(obs), _p(hidden), _p(pi), _p(v))
        return hidden, 0.0, pi, float(v[0])

    def recurrent_inference(self, hidden, action):
        hidden = _f32(hidden).reshape(-1)
        out = np.zeros(self.hidden_size, np.float32)
        pi = np.zeros(self.A, np.float32)
        r = np.zeros(1, np.float32)
        v = np.zeros(1, np.float32)
        lib().mzo_recurrent_inference(self.h, _p(hidden), int(action), _p(out), _p(r), _p(pi), _p(v))
        return out, float(r[0]), pi, float(v[0])


def uct_search(cfg, net, obs, mask, current_player, opponent_player, temperature, deterministic=False, noise=None, u_tie=None,
               u_final=0.5, want_trace=False):
    """mcts.py:302-407 with injected randomness.  Returns dict(action, pi, root_value, visits, ...)."""
    A, S = cfg.num_actions, cfg.num_simulations
    obs = _f32(obs).reshape(-1)
    mask_a = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    noise_a = None if noise is None else np.ascontiguousarray(noise, dtype=np.float64)
    u_tie_a = np.ascontiguousarray(u_tie if u_tie is not None else np.full(4 * S + 8, 0.5), dtype=np.float64)
    rng = RngInputs(noise_a.ctypes.data if noise_a is not None else None, u_tie_a.ctypes.data, len(u_tie_a), float(u_final))
    pi = np.zeros(A, np.float64)
    visits = np.zeros(A, np.int32)
    tp = np.zeros(S, np.int32)
    ta = np.zeros(S, np.int32)
    mm = np.zeros(2, np.float64)
    res = lib().mzo_uct_search(
        C.byref(cfg), net.h, _p(obs), _p(mask_a), int(current_player), int(opponent_player), float(temperature), int(bool(deterministic)),
        C.byref(rng), _p(pi), _p(visits), _p(tp), _p(ta), _p(mm),
    )
    if res.status != 0:
        raise RuntimeError(f'oracle search failed: status {res.status}')
    return dict(action=res.action, pi=pi, root_value=res.root_value, visits=visits, trace_parent=tp, trace_action=ta, minmax=mm,
                n_tie_used=res.n_tie_used)


def uct_search_batch(cfg, net, obs, mask, cur_player, opp_player, temperature, deterministic=False, noise=None, u_tie=None, u_final=None,
                     num_threads=0):
    A, S = cfg.num_actions, cfg.num_simulations
    B = obs.shape[0]
    obs = _f32(obs).reshape(B, -1)
    mask_a = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    noise_a = None if noise is None else np.ascontiguousarray(noise, dtype=np.float64)
    u_tie_a = np.ascontiguousarray(u_tie if u_tie is not None else np.full((B, 4 * S + 8), 0.5), dtype=np.float64)
    u_final_a = np.ascontiguousarray(u_final if u_final is not None else np.full(B, 0.5), dtype=np.float64)
    cur = np.ascontiguousarray(np.broadcast_to(cur_player, (B,)), dtype=np.int32)
    opp = np.ascontiguousarray(np.broadcast_to(opp_player, (B,)), dtype=np.int32)
    temp = np.ascontiguousarray(np.broadcast_to(temperature, (B,)), dtype=np.float64)
    action = np.zeros(B, np.int32)
    pi = np.zeros((B, A), np.float64)
    root = np.zeros(B, np.float64)
    visits = np.zeros((B, A), np.int32)
    st = lib().mzo_uct_search_batch(
        C.byref(cfg), net.h, B, _p(obs), _p(mask_a), _p(cur), _p(opp), _p(temp), int(bool(deterministic)), _p(noise_a), _p(u_tie_a),
        u_tie_a.shape[1], _p(u_final_a), _p(action), _p(pi), _p(root), _p(visits), int(num_threads),
    )
    if st != 0:
        raise RuntimeError(f'oracle batch search failed: status {st}')
    return dict(action=action, pi=pi, root_value=root, visits=visits)


def prepare_root_prior(pi0, noise, eps, mask, deterministic):
    pi0 = _f32(pi0)
    A = len(pi0)
    p64 = np.zeros(A, np.float64)
    p32 = np.zeros(A, np.float32)
    noise_a = None if noise is None else np.ascontiguousarray(noise, np.float64)
    mask_a = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    lib().mzo_prepare_root_prior(_p(pi0), A, _p(noise_a), float(eps), _p(mask_a), int(bool(deterministic)), _p(p64), _p(p32))
    return p64, p32


def generate_play_policy(visits, temperature):
    v = np.ascontiguousarray(visits, np.int32)
    pi = np.zeros(len(v), np.float64)
    lib().mzo_generate_play_policy(_p(v), len(v), float(temperature), _p(pi))
    return pi


def sample_action(pi, u):
    pi = np.ascontiguousarray(pi, np.float64)
    return lib().mzo_sample_action(_p(pi), len(pi), float(u))


def signed_parabolic(x):
    x = _f32(x)
    return np.array([lib().mzo_signed_parabolic(float(v)) for v in x.reshape(-1)], np.float32).reshape(x.shape)


def logits_to_value(logits):
    logits = _f32(logits)
    return np.array([lib().mzo_logits_to_value(_p(row), logits.shape[-1]) for row in logits.reshape(-1, logits.shape[-1])], np.float32)


def normalize_hidden(h):
    """h: [C] or [C, H, W] (batch of one); min/max over channels."""
    h = _f32(h).copy()
    c = h.shape[0]
    lib().mzo_normalize_hidden(_p(h), c, int(h.size // c))
    return h


def n_step_target(rewards, root_values, td_steps, discount):
    r = np.ascontiguousarray(rewards, np.float64)
    v = np.ascontiguousarray(root_values, np.float64)
    out = np.zeros(len(r), np.float64)
    lib().mzo_n_step_target(_p(r), _p(v), len(r), int(td_steps), float(discount), _p(out))
    return out


def mc_return_target(rewards, player_ids):
    r = np.ascontiguousarray(rewards, np.float64)
    p = np.ascontiguousarray(player_ids, np.int32)
    out = np.zeros(len(r), np.float64)
    lib().mzo_mc_return_target(_p(r), _p(p), len(r), _p(out))
    return out


def make_unroll_sequence(observations, actions, rewards, pi_probs, values, priorities, unroll_steps):
    """pipeline.py:710-767 as arrays: pad K absorbing steps (action 0, reward 0, value 0, uniform policy) and slide a
    window of K.  Returns (state[T,...], action[T,K] int8, reward[T,K] f32, value[T,K] f32, pi[T,K,A] f32, priority[T])."""
    T = len(observations)
    K = unroll_steps
    A = np.asarray(pi_probs[0]).shape[0]
    act = np.concatenate([np.asarray(actions, np.int64), np.zeros(K, np.int64)])
    rew = np.concatenate([np.asarray(rewards, np.float64), np.zeros(K)])
    val = np.concatenate([np.asarray(values, np.float64), np.zeros(K)])
    pis = np.concatenate([np.asarray(pi_probs, np.float64), np.full((K, A), 1.0 / A)])
    idx = np.arange(T)[:, None] + np.arange(K)[None, :]
    return (np.stack([np.asarray(o) for o in observations]), act[idx].astype(np.int8), rew[idx].astype(np.float32),
            val[idx].astype(np.float32), pis[idx].astype(np.float32), np.asarray(priorities, np.float64)[:T])


class BoardEnv:
    """BoardGameEnv / TicTacToeEnv / GomokuEnv semantics (games/env.py)."""

    def __init__(self, board_size=3, stack=4, num_to_win=3):
        self.b = Board()
        self.board_size, self.stack, self.num_to_win = board_size, stack, num_to_win
        self.reset()

    def reset(self):
        lib().mzo_board_reset(C.byref(self.b), self.board_size, self.stack, self.num_to_win)
        return self.observation()

    def observation(self):
        n = self.board_size
        obs = np.zeros((2 * self.stack + 1, n, n), np.int8)
        lib().mzo_board_observation(C.byref(self.b), _p(obs))
        return obs

    def step(self, action):
        r = C.c_double(0.0)
        d = C.c_int32(0)
        st = lib().mzo_board_step(C.byref(self.b), int(action), C.byref(r), C.byref(d))
        if st != 0:
            raise ValueError(f'invalid board step: {st}')
        return self.observation(), r.value, bool(d.value)

    @property
    def actions_mask(self):
        return np.array(self.b.mask[: self.b.num_actions], dtype=np.uint8).astype(bool)

    @property
    def current_player(self):
        return self.b.current_player

    @property
    def opponent_player(self):
        return 3 - self.b.current_player

    @property
    def winner(self):
        return self.b.winner


class CartPoleEnv:
    """CartPole-v1 + StackFrameAndAction(stack) + PlayerIdAndActionMaskWrapper (gym_env.py:271-365,436-459)."""

    def __init__(self, stack=4):
        self.e = CartPole()
        self.stack = stack
        self.stacked = np.zeros((stack, 5), np.float32)

    def reset(self, init):
        init = np.ascontiguousarray(init, np.float64)
        lib().mzo_cartpole_reset(C.byref(self.e), _p(init))
        obs = init.astype(np.float32)
        lib().mzo_stack_reset(_p(self.stacked), self.stack, 4, _p(obs), 2)
        return self.stacked.copy()

    def step(self, action):
        obs = np.zeros(4, np.float32)
        r = lib().mzo_cartpole_step(C.byref(self.e), int(action), _p(obs))
        lib().mzo_stack_push(_p(self.stacked), self.stack, 4, _p(obs), int(action), 2)
        return self.stacked.copy(), r, bool(self.e.done)
