#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE implementation (imported read-only from
/root/reference) and records inputs / injected randomness / outputs as small .npz fixtures
under tests/golden/.

Runs ONLY in the build container (the GPU box has no /root/reference).  The fixtures are data:
inputs, the random draws the reference consumed, and the reference's outputs.  No reference
source text is stored.

    python oracle/gen_golden.py            # regenerate everything
    python oracle/gen_golden.py tree nets  # regenerate selected groups

Groups:
  tree   G1  reference uct_search (mcts.py:302-407) driven by a scripted fake network
  nets   G2  MuZeroMLPNet / MuZeroBoardGameNet / MuZeroAtariNet initial+recurrent inference
  search G3  end-to-end uct_search with real (seeded random-weight) networks
  ckpt   G3  the same on the SHIPPED trained checkpoints (saved_checkpoints/*): inputs, recorded draws and the reference's
             outputs only -- the weights never leave /root/reference, so the test that replays these through the oracle
             (tests/test_oracle_ckpt.py) runs in the build container only
  legacy G3' deterministic (evaluator) searches with the scalar promotion of numpy 1.21 -- the version the reference pins
             (requirements.txt:21) -- emulated exactly under this container's numpy 2: child priors stored as float64 scalars, so that
             mcts.py:189-197 multiplies in float64 and rounds once (tests/golden/legacy_cases.npz; oracle / kernels: legacy_scalar_promotion)
  pipe   G4  pipeline/util/mcts helper functions (incl. the reference's own KATs)
  env    G5  TicTacToe / Gomoku scripted games (incl. the reference tests' win lines)
  learn  L   PrioritizedReplay sampling, calc_loss (loss, priorities, gradients), 3 optimizer steps, 2-hot projection
  play   G6  one full reference run_self_play episode on TicTacToe
  classic G7 the classic-control observation path: StackFrameAndAction + PlayerIdAndActionMaskWrapper
             (gym_env.py:271-365) over a scripted base env, and reference run_self_play episodes on it with a small
             acc_seq_length, i.e. through the mid-episode flush branch (pipeline.py:118-142)
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

_refshim.install()

import torch  # noqa: E402

import muzero.mcts as ref_mcts  # noqa: E402
import muzero.network as ref_network  # noqa: E402
import muzero.util as ref_util  # noqa: E402
import muzero.config as ref_config  # noqa: E402
import muzero.pipeline as ref_pipeline  # noqa: E402
from muzero.games.tictactoe import TicTacToeEnv  # noqa: E402
from muzero.games.gomoku import GomokuEnv  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
torch.set_num_threads(1)


# --------------------------------------------------------------------------------------------
# Recording of the global-numpy-RNG draws consumed by the reference (mcts.py:124,245,404).
# The reference arithmetic is untouched: the real numpy functions are called and their results
# returned; we only log what they produced.
# --------------------------------------------------------------------------------------------
class DrawRecorder:
    def __init__(self):
        self.noise = None
        self.tie_idx = []  # index *within the candidate list* for every tie-break with >1 candidate
        self.tie_n = []
        self.final_u = np.nan
        self.final_action = -1
        self.visits = None
        self._orig = {}

    def __enter__(self):
        self._orig = dict(dirichlet=np.random.dirichlet, choice=np.random.choice, gpp=ref_mcts.generate_play_policy)
        rec = self

        def dirichlet(alphas, *a, **k):
            out = rec._orig['dirichlet'](alphas, *a, **k)
            rec.noise = np.array(out, dtype=np.float64)
            return out

        def choice(a, *args, **kwargs):
            p = kwargs.get('p', None)
            if p is None:
                cand = np.asarray(a)
                res = rec._orig['choice'](a, *args, **kwargs)
                if cand.shape[0] > 1:
                    rec.tie_idx.append(int(np.where(cand == res)[0][0]))
                    rec.tie_n.append(int(cand.shape[0]))
                return res
            # final sample: recover the uniform double numpy consumed (one random_sample()).
            st = np.random.get_state()
            res = rec._orig['choice'](a, *args, **kwargs)
            st_after = np.random.get_state()
            np.random.set_state(st)
            u = np.random.random_sample()
            assert all(np.array_equal(x, y) for x, y in zip(np.random.get_state()[1:2], st_after[1:2]))
            np.random.set_state(st_after)
            rec.final_u = float(u)
            rec.final_action = int(res)
            return res

        def gpp(visits_count, temperature):
            rec.visits = np.array(visits_count, dtype=np.int64)
            return rec._orig['gpp'](visits_count, temperature)

        np.random.dirichlet = dirichlet
        np.random.choice = choice
        ref_mcts.generate_play_policy = gpp
        return self

    def __exit__(self, *exc):
        np.random.dirichlet = self._orig['dirichlet']
        np.random.choice = self._orig['choice']
        ref_mcts.generate_play_policy = self._orig['gpp']

    def as_dict(self, A, max_ties):
        assert len(self.tie_idx) <= max_ties, (len(self.tie_idx), max_ties)
        u_tie = np.full((max_ties,), 0.5, dtype=np.float64)
        for i, (idx, n) in enumerate(zip(self.tie_idx, self.tie_n)):
            u_tie[i] = (idx + 0.5) / n
        return dict(
            noise=self.noise if self.noise is not None else np.zeros((A,), np.float64),
            has_noise=np.int32(self.noise is not None),
            u_tie=u_tie,
            n_tie=np.int32(len(self.tie_idx)),
            u_final=np.float64(self.final_u if np.isfinite(self.final_u) else 0.5),
            visits=self.visits.astype(np.int32),
        )


def make_config(discount, alpha, sims, board, bounds, value_support=1, reward_support=1):
    return ref_config.MuZeroConfig(
        discount=discount,
        dirichlet_alpha=alpha,
        num_simulations=sims,
        batch_size=8,
        td_steps=10,
        lr_init=0.01,
        lr_milestones=[10],
        visit_softmax_temperature_fn=lambda a, b: 1.0,
        known_bounds=ref_config.KnownBounds(*bounds) if bounds is not None else None,
        value_support_size=value_support,
        reward_support_size=reward_support,
        is_board_game=board,
    )


def cfg_arrays(cfg):
    kb = cfg.known_bounds
    return dict(
        discount=np.float64(cfg.discount),
        alpha=np.float64(cfg.root_dirichlet_alpha),
        eps=np.float64(cfg.root_exploration_eps),
        sims=np.int32(cfg.num_simulations),
        board=np.int32(cfg.is_board_game),
        has_bounds=np.int32(kb is not None),
        kb_min=np.float64(kb.min if kb else 0.0),
        kb_max=np.float64(kb.max if kb else 0.0),
        pb_c_base=np.float64(cfg.pb_c_base),
        pb_c_init=np.float64(cfg.pb_c_init),
    )


# --------------------------------------------------------------------------------------------
# G1: tree-only.  A fake network scripts (value, reward) per simulation and labels every node's
# hidden state with its creation index, so the (parent, action) chosen by every simulation can
# be read back from the arguments of recurrent_inference.
# --------------------------------------------------------------------------------------------
class ScriptedNet:
    def __init__(self, pi0, values, rewards):
        self.pi0 = pi0.astype(np.float32)
        self.values = values
        self.rewards = rewards
        self.calls = 0
        self.trace_parent = []
        self.trace_action = []

    def initial_inference(self, x):
        return ref_network.NetworkOutputs(
            hidden_state=np.array([0.0], dtype=np.float32), reward=0.0, pi_probs=self.pi0.copy(), value=0.123
        )

    def recurrent_inference(self, hidden_state, action):
        s = self.calls
        self.calls += 1
        self.trace_parent.append(int(hidden_state.reshape(-1)[0].item()))
        self.trace_action.append(int(action.reshape(-1)[0].item()))
        # value / reward arrive in the tree as Python floats holding float32 values (network.py:107-108).
        return ref_network.NetworkOutputs(
            hidden_state=np.array([float(s + 1)], dtype=np.float32),
            reward=float(np.float32(self.rewards[s])),
            pi_probs=self.pi0.copy(),
            value=float(np.float32(self.values[s])),
        )


def gen_tree():
    rng = np.random.RandomState(1234)
    cases = []
    spec = [
        # A, S, board, bounds, discount, alpha, deterministic, temperature, players, mask_kind, value_kind
        (2, 50, False, None, 0.997, 0.25, False, 1.0, (1, 1), 'all', 'positive'),
        (2, 50, False, None, 0.997, 0.25, False, 0.5, (1, 1), 'all', 'positive'),
        (2, 50, False, None, 0.997, 0.25, True, 0.25, (1, 1), 'all', 'positive'),
        (2, 50, False, None, 0.997, 0.25, False, 0.0, (1, 1), 'all', 'mixed'),
        (4, 50, False, None, 0.997, 0.25, False, 1.0, (1, 1), 'all', 'mixed'),
        (4, 30, False, None, 0.997, 0.25, True, 1.0, (1, 1), 'random', 'mixed'),
        (6, 30, False, None, 0.997, 0.25, False, 0.25, (1, 1), 'all', 'positive'),
        (10, 25, True, (-1, 1), 1.0, 0.25, False, 1.0, (1, 2), 'all', 'unit'),
        (10, 25, True, (-1, 1), 1.0, 0.25, False, 0.1, (2, 1), 'random', 'unit'),
        (10, 25, True, (-1, 1), 1.0, 0.25, True, 0.1, (1, 2), 'random', 'unit'),
        (10, 25, True, (-1, 1), 1.0, 0.25, False, 1.0, (1, 2), 'random', 'big'),  # values escape the known bounds
        (10, 25, True, None, 1.0, 0.25, False, 1.0, (2, 1), 'random', 'unit'),
        (10, 40, True, (-1, 1), 1.0, 0.0, False, 1.0, (1, 2), 'all', 'zero'),  # alpha=0: no noise, all-zero values => many ties
        (10, 12, False, None, 0.9, 0.3, False, 0.3, (1, 1), 'random', 'mixed'),  # non-integer exponent 1/T
        (3, 1, False, None, 0.997, 0.25, False, 1.0, (1, 1), 'all', 'mixed'),  # single simulation
        (10, 40, True, (-1, 1), 1.0, 0.0, False, 1.0, (1, 2), 'all', 'zero_uniform'),  # uniform prior + zero values: ties at every level
        (4, 30, False, None, 0.997, 0.0, True, 1.0, (1, 1), 'all', 'zero_uniform'),  # same, deterministic (float32 prior path)
        (10, 25, True, (-1, 1), 1.0, 0.25, True, 0.0, (1, 2), 'random', 'zero_uniform'),
        (226, 60, True, (-1, 1), 1.0, 0.03, False, 1.0, (1, 2), 'random', 'unit'),
        (226, 40, True, (-1, 1), 1.0, 0.03, True, 0.1, (2, 1), 'all', 'unit'),
        (82, 30, True, (-1, 1), 1.0, 0.03, False, 0.1, (1, 2), 'random', 'unit'),
    ]
    # add a few randomised repeats of the two benchmark shapes
    for r in range(6):
        spec.append((2, 50, False, None, 0.997, 0.25, False, 1.0, (1, 1), 'all', 'positive'))
        spec.append((10, 25, True, (-1, 1), 1.0, 0.25, False, 1.0 if r % 2 else 0.1, (1 + r % 2, 2 - r % 2), 'random', 'unit'))

    out = {}
    for ci, (A, S, board, bounds, disc, alpha, det, T, players, mask_kind, vkind) in enumerate(spec):
        cfg = make_config(disc, alpha, S, board, bounds)
        logits = rng.randn(A).astype(np.float32) * 1.5
        if vkind == 'zero_uniform':
            logits[:] = 0.0
        pi0 = torch.softmax(torch.from_numpy(logits), dim=0).numpy()
        if vkind == 'positive':
            values = rng.uniform(0.0, 12.0, size=S)
            rewards = rng.uniform(0.0, 1.5, size=S)
        elif vkind == 'mixed':
            values = rng.uniform(-5.0, 5.0, size=S)
            rewards = rng.uniform(-1.0, 1.0, size=S)
        elif vkind == 'unit':
            values = rng.uniform(-1.0, 1.0, size=S)
            rewards = np.where(rng.rand(S) < 0.2, rng.uniform(-1, 1, size=S), 0.0)
        elif vkind == 'big':
            values = rng.uniform(-3.0, 3.0, size=S)
            rewards = rng.uniform(-2.0, 2.0, size=S)
        else:
            values = np.zeros(S)
            rewards = np.zeros(S)
        values = values.astype(np.float32)
        rewards = rewards.astype(np.float32)
        if mask_kind == 'all':
            mask = np.ones(A, dtype=bool)
        else:
            mask = rng.rand(A) < 0.6
            mask[rng.randint(A)] = True
            if mask.sum() < 2 and A > 1:
                mask[(np.argmax(mask) + 1) % A] = True
        net = ScriptedNet(pi0, values, rewards)
        np.random.seed(1000 + ci)
        with DrawRecorder() as rec:
            action, pi, root_value = ref_mcts.uct_search(
                state=np.zeros((1,), np.float32),
                network=net,
                device=torch.device('cpu'),
                config=cfg,
                temperature=float(T),
                actions_mask=mask,
                current_player=players[0],
                opponent_player=players[1],
                deterministic=det,
            )
        d = dict(
            A=np.int32(A),
            pi0=pi0,
            values=values,
            rewards=rewards,
            mask=mask.astype(np.uint8),
            temperature=np.float64(T),
            deterministic=np.int32(det),
            cur_player=np.int32(players[0]),
            opp_player=np.int32(players[1]),
            trace_parent=np.array(net.trace_parent, np.int32),
            trace_action=np.array(net.trace_action, np.int32),
            out_action=np.int32(action),
            out_pi=np.asarray(pi, np.float64),
            out_root_value=np.float64(root_value),
        )
        d.update(cfg_arrays(cfg))
        d.update(rec.as_dict(A, max_ties=4 * S + 8))
        for k, v in d.items():
            out[f'c{ci}_{k}'] = v
        cases.append(ci)
    out['num_cases'] = np.int32(len(cases))
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'tree_cases.npz'), **out)
    print(f'tree: {len(cases)} cases')


# --------------------------------------------------------------------------------------------
# Seeded weights: fixtures only store the seed; tests/helpers.py regenerates identical weights.
# --------------------------------------------------------------------------------------------
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tests'))
from helpers import MLP_CASES, CONV_CASES, seeded_state_dict  # noqa: E402
import helpers  # noqa: E402


def _infer_case(net, obs, actions, prefix, out):
    """Run reference initial_inference on obs then a chain of recurrent_inference with `actions`."""
    with torch.no_grad():
        o = net.initial_inference(torch.from_numpy(obs).to(torch.float32)[None, ...])
    out[f'{prefix}_obs'] = obs
    out[f'{prefix}_init_hidden'] = np.asarray(o.hidden_state, np.float32)
    out[f'{prefix}_init_pi'] = np.asarray(o.pi_probs, np.float32)
    out[f'{prefix}_init_value'] = np.float32(o.value)
    out[f'{prefix}_init_reward'] = np.float32(o.reward)
    h = o.hidden_state
    hs, rs_, vs, ps = [], [], [], []
    for a in actions:
        with torch.no_grad():
            o = net.recurrent_inference(torch.from_numpy(h)[None, ...], torch.tensor([[int(a)]], dtype=torch.long))
        h = o.hidden_state
        hs.append(np.asarray(h, np.float32))
        rs_.append(np.float32(o.reward))
        vs.append(np.float32(o.value))
        ps.append(np.asarray(o.pi_probs, np.float32))
    out[f'{prefix}_actions'] = np.asarray(actions, np.int32)
    out[f'{prefix}_rec_hidden'] = np.stack(hs)
    out[f'{prefix}_rec_reward'] = np.asarray(rs_, np.float32)
    out[f'{prefix}_rec_value'] = np.asarray(vs, np.float32)
    out[f'{prefix}_rec_pi'] = np.stack(ps)


def build_mlp(case):
    return helpers.build_mlp(case, ref_network)


def build_conv(case):
    return helpers.build_conv(case, ref_network)


def gen_nets():
    rng = np.random.RandomState(77)
    out = {}
    for case in MLP_CASES:
        name, ishape, A = case[0], case[1], case[2]
        net = build_mlp(case)
        for j in range(3):
            obs = rng.uniform(-1.0, 1.0, size=ishape).astype(np.float32)
            actions = rng.randint(0, A, size=6)
            _infer_case(net, obs, actions, f'mlp_{name}_{j}', out)
    for case in CONV_CASES:
        name, kind, ishape, A = case[0], case[1], case[2], case[3]
        net = build_conv(case)
        for j in range(2):
            if kind == 'board':
                obs = (rng.rand(*ishape) < 0.3).astype(np.float32)
            else:
                obs = rng.uniform(0.0, 255.0, size=ishape).astype(np.float32)
            actions = rng.randint(0, A, size=3)
            _infer_case(net, obs, actions, f'conv_{name}_{j}', out)
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'net_cases.npz'), **out)
    print('nets: done,', len(out), 'arrays')


# --------------------------------------------------------------------------------------------
# G3: end-to-end search with real networks.
# --------------------------------------------------------------------------------------------
def _search_case(net, cfg, obs, mask, players, T, det, seed, prefix, out, A):
    np.random.seed(seed)
    with DrawRecorder() as rec:
        action, pi, root_value = ref_mcts.uct_search(
            state=obs,
            network=net,
            device=torch.device('cpu'),
            config=cfg,
            temperature=float(T),
            actions_mask=mask,
            current_player=players[0],
            opponent_player=players[1],
            deterministic=det,
        )
    d = dict(
        obs=obs,
        mask=mask.astype(np.uint8),
        temperature=np.float64(T),
        deterministic=np.int32(det),
        cur_player=np.int32(players[0]),
        opp_player=np.int32(players[1]),
        out_action=np.int32(action),
        out_pi=np.asarray(pi, np.float64),
        out_root_value=np.float64(root_value),
        # the global numpy stream right after the search (seeded with `seed` before it): a literal "identical seeds" run must leave the
        # generator in the same place -- same number of words consumed by the Dirichlet draw, every tie-break and the final sample
        seed=np.int64(seed),
        next_uniform=np.float64(np.random.random_sample()),
    )
    d.update(rec.as_dict(A, max_ties=4 * cfg.num_simulations + 8))
    for k, v in d.items():
        out[f'{prefix}_{k}'] = v


def gen_search():
    rng = np.random.RandomState(99)
    out = {}
    # C2 shape: CartPole MLP (seeded random weights, full size)
    case = MLP_CASES[0]
    net = build_mlp(case)
    cfg = make_config(0.997, 0.25, 50, False, None, 31, 31)
    for k, v in cfg_arrays(cfg).items():
        out[f'cartpole_{k}'] = v
    n = 0
    for j in range(12):
        obs = rng.uniform(-0.5, 0.5, size=(4, 5)).astype(np.float32)
        obs[:, 4] = (rng.randint(0, 2, size=4) + 1) / 2.0
        T = [1.0, 0.5, 0.25][j % 3]
        det = (j % 4 == 3)
        _search_case(net, cfg, obs, np.ones(2, bool), (1, 1), T, det, 500 + j, f'cartpole_{j}', out, 2)
        n += 1
    out['cartpole_n'] = np.int32(n)
    # C3 shape: TicTacToe MLP
    case = MLP_CASES[2]
    net = build_mlp(case)
    cfg = make_config(1.0, 0.25, 25, True, (-1, 1), 1, 1)
    for k, v in cfg_arrays(cfg).items():
        out[f'tictactoe_{k}'] = v
    n = 0
    for j in range(12):
        env = TicTacToeEnv()
        obs = env.reset()
        for _ in range(rng.randint(0, 5)):
            legal = np.where(env.actions_mask[:9])[0]
            obs, _, done, _ = env.step(int(rng.choice(legal)))
            if done:
                break
        if env.is_game_over:
            env = TicTacToeEnv()
            obs = env.reset()
        T = 1.0 if j % 2 == 0 else 0.1
        det = (j % 5 == 4)
        _search_case(
            net, cfg, obs.astype(np.int8), env.actions_mask.copy(), (env.current_player, env.opponent_player), T, det, 700 + j,
            f'tictactoe_{j}', out, 10,
        )
        n += 1
    out['tictactoe_n'] = np.int32(n)
    # LunarLander-shaped 4-action MLP
    case = MLP_CASES[1]
    net = build_mlp(case)
    cfg = make_config(0.997, 0.25, 50, False, None, 31, 31)
    for k, v in cfg_arrays(cfg).items():
        out[f'lunar_{k}'] = v
    for j in range(4):
        obs = rng.uniform(-1, 1, size=(4, 9)).astype(np.float32)
        _search_case(net, cfg, obs, np.ones(4, bool), (1, 1), 1.0, False, 800 + j, f'lunar_{j}', out, 4)
    out['lunar_n'] = np.int32(4)
    # small conv board net (TicTacToe --nouse_mlp_net shape) and small atari net
    case = CONV_CASES[0]
    net = build_conv(case)
    cfg = make_config(1.0, 0.25, 25, True, (-1, 1), 1, 1)
    for k, v in cfg_arrays(cfg).items():
        out[f'board3_{k}'] = v
    for j in range(4):
        env = TicTacToeEnv()
        obs = env.reset()
        for _ in range(j):
            legal = np.where(env.actions_mask[:9])[0]
            obs, _, done, _ = env.step(int(rng.choice(legal)))
        _search_case(
            net, cfg, obs.astype(np.int8), env.actions_mask.copy(), (env.current_player, env.opponent_player), 1.0, False, 900 + j,
            f'board3_{j}', out, 10,
        )
    out['board3_n'] = np.int32(4)
    case = CONV_CASES[3]
    net = build_conv(case)
    cfg = make_config(0.997, 0.25, 12, False, None, 11, 11)
    for k, v in cfg_arrays(cfg).items():
        out[f'atari_s_{k}'] = v
    for j in range(2):
        obs = rng.uniform(0, 255, size=(4, 96, 96)).astype(np.float32)
        _search_case(net, cfg, obs, np.ones(6, bool), (1, 1), 1.0, False, 950 + j, f'atari_s_{j}', out, 6)
    out['atari_s_n'] = np.int32(2)
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'search_cases.npz'), **out)
    print('search: done')



# --------------------------------------------------------------------------------------------
# G3 on the shipped checkpoints (pipeline.py:810-817 loads them the same way).  Trained networks are where
# near-ties live: value ranges are realistic (CartPole values ~ tens, TicTacToe in [-1, 1]) instead of the degenerate ones of
# random weights.  Only inputs / draws / outputs are stored.
# --------------------------------------------------------------------------------------------
CKPT_DIR = '/root/reference/saved_checkpoints'
CKPT_CASES = [  # name, file, input shape, actions, planes, support, (discount, sims, board, bounds)
    ('cartpole', 'CartPole-v1_train_steps_44800', (4, 5), 2, 512, 31, (0.997, 50, False, None)),
    ('lunar', 'LunarLander-v2_train_steps_58400', (4, 9), 4, 512, 31, (0.997, 50, False, None)),
    ('tictactoe', 'TicTacToe_train_steps_35000', (9, 3, 3), 10, 256, 1, (1.0, 25, True, (-1, 1))),
]


def gen_ckpt():
    out = {}
    rng = np.random.RandomState(4242)
    for name, fname, ishape, A, P, sup, (disc, sims, board, bounds) in CKPT_CASES:
        net = ref_network.MuZeroMLPNet(ishape, A, P, sup, sup, 64)
        ck = torch.load(os.path.join(CKPT_DIR, fname), map_location='cpu', weights_only=False)
        net.load_state_dict(ck['network'])
        net.eval()
        cfg = make_config(disc, 0.25, sims, board, bounds, sup, sup)
        for k, v in cfg_arrays(cfg).items():
            out[f'{name}_{k}'] = v
        n = 0
        for j in range(16):
            if board:
                env = TicTacToeEnv()
                obs = env.reset()
                for _ in range(rng.randint(0, 6)):
                    legal = np.where(env.actions_mask[:9])[0]
                    obs, _, done, _ = env.step(int(rng.choice(legal)))
                    if done:
                        break
                if env.is_game_over:
                    env = TicTacToeEnv()
                    obs = env.reset()
                obs, mask, players = obs.astype(np.int8), env.actions_mask.copy(), (env.current_player, env.opponent_player)
            else:
                D = ishape[1] - 1
                scale = 0.2 if name == 'cartpole' else 1.0  # CartPole states near the upright pole; LunarLander's 8 features ~ U(-1, 1)
                obs = rng.uniform(-scale, scale, size=ishape).astype(np.float32)
                obs[:, D] = (rng.randint(0, A, size=ishape[0]) + 1) / float(A)  # the action plane of StackFrameAndAction (gym_env.py:319-322)
                mask, players = np.ones(A, bool), (1, 1)
            for seed in range(2):
                T = [1.0, 0.25][seed] if not board else [1.0, 0.1][seed]
                det = (j % 4 == 3) and seed == 1
                _search_case(net, cfg, obs, mask, players, T, det, 3000 + 10 * j + seed, f'{name}_{n}', out, A)
                n += 1
        out[f'{name}_n'] = np.int32(n)
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'ckpt_cases.npz'), **out)
    print('ckpt: done,', len(out), 'arrays')


# --------------------------------------------------------------------------------------------
# G4: helpers
# --------------------------------------------------------------------------------------------
def gen_pipe():
    rng = np.random.RandomState(5)
    out = {}
    # the reference's own KATs (tests/pipeline_test.py:24-53)
    out['nstep_kat1_rewards'] = np.ones(5)
    out['nstep_kat1_roots'] = np.zeros(5)
    out['nstep_kat1_out'] = np.array(ref_pipeline.compute_n_step_target([1.0] * 5, [0] * 5, 5, 0.997))
    rv = [0.1 * (i + 1) for i in range(10)]
    out['nstep_kat2_rewards'] = np.ones(10)
    out['nstep_kat2_roots'] = np.array(rv)
    out['nstep_kat2_out'] = np.array(ref_pipeline.compute_n_step_target([1.0] * 10, rv, 5, 0.997))
    for j, (T, td, disc) in enumerate([(1, 10, 0.997), (7, 10, 0.997), (37, 10, 0.997), (23, 5, 0.9), (12, 0, 1.0), (200, 10, 0.997)]):
        r = rng.uniform(-1, 2, size=T).astype(np.float32).astype(np.float64)
        v = rng.uniform(-3, 30, size=T)
        out[f'nstep_{j}_rewards'] = r
        out[f'nstep_{j}_roots'] = v
        out[f'nstep_{j}_td'] = np.int32(td)
        out[f'nstep_{j}_discount'] = np.float64(disc)
        out[f'nstep_{j}_out'] = np.array(ref_pipeline.compute_n_step_target(list(r), list(v), td, disc))
    out['nstep_n'] = np.int32(6)
    # mc return
    mc = [
        ([0, 0, 0, 0, 1.0], [1, 2, 1, 2, 1]),
        ([0, 0, 0, -1.0], [1, 2, 1, 2]),
        ([0, 0, 0, 0, 0, 0, 0, 0, 0.0], [1, 2, 1, 2, 1, 2, 1, 2, 1]),
        ([1.0], [2]),
        ([0, 0, 0, 0, 0, 1.0], [1, 2, 1, 2, 1, 2]),
    ]
    for j, (r, p) in enumerate(mc):
        out[f'mc_{j}_rewards'] = np.array(r, np.float64)
        out[f'mc_{j}_players'] = np.array(p, np.int32)
        out[f'mc_{j}_out'] = np.array(ref_pipeline.compute_mc_return_target(list(r), list(p)), np.float64)
    out['mc_n'] = np.int32(len(mc))
    # make_unroll_sequence
    for j, (T, A, oshape) in enumerate([(1, 2, (4, 5)), (4, 2, (4, 5)), (9, 10, (9, 3, 3)), (13, 4, (4, 9))]):
        obs = [rng.uniform(-1, 1, size=oshape).astype(np.float32) for _ in range(T)]
        actions = [int(x) for x in rng.randint(0, A, size=T)]
        rewards = [float(x) for x in rng.uniform(-1, 1, size=T)]
        pis = []
        for _ in range(T):
            p = rng.rand(A)
            pis.append(p / p.sum())
        values = [float(x) for x in rng.uniform(-2, 2, size=T)]
        prios = np.abs(rng.randn(T))
        out[f'unroll_{j}_obs'] = np.stack(obs)
        out[f'unroll_{j}_actions'] = np.array(actions, np.int32)
        out[f'unroll_{j}_rewards'] = np.array(rewards, np.float64)
        out[f'unroll_{j}_pis'] = np.stack(pis)
        out[f'unroll_{j}_values'] = np.array(values, np.float64)
        out[f'unroll_{j}_prios'] = prios
        seq = list(ref_pipeline.make_unroll_sequence(list(obs), list(actions), list(rewards), list(pis), list(values), prios, 5))
        out[f'unroll_{j}_out_state'] = np.stack([t.state for t, _ in seq])
        out[f'unroll_{j}_out_action'] = np.stack([t.action for t, _ in seq])
        out[f'unroll_{j}_out_reward'] = np.stack([t.reward for t, _ in seq])
        out[f'unroll_{j}_out_value'] = np.stack([t.value for t, _ in seq])
        out[f'unroll_{j}_out_pi'] = np.stack([t.pi_prob for t, _ in seq])
        out[f'unroll_{j}_out_prio'] = np.array([p for _, p in seq], np.float64)
    out['unroll_n'] = np.int32(4)
    # generate_play_policy
    j = 0
    for A in (2, 4, 10, 226):
        for T in (0.0, 0.1, 0.25, 0.3, 0.5, 0.7, 1.0):
            v = rng.randint(0, 60, size=A).astype(np.int32)
            v[rng.randint(A)] += 1
            out[f'policy_{j}_visits'] = v
            out[f'policy_{j}_T'] = np.float64(T)
            out[f'policy_{j}_out'] = ref_mcts.generate_play_policy(v, float(T))
            j += 1
    out['policy_n'] = np.int32(j)
    # add_dirichlet_noise + set_illegal_action_probs_to_zero (f32 prior -> f64 noised prior)
    j = 0
    for A in (2, 4, 10, 82, 226):
        for alpha in (0.25, 0.03):
            p = torch.softmax(torch.from_numpy(rng.randn(A).astype(np.float32)), 0).numpy()
            np.random.seed(40 + j)
            with DrawRecorder() as rec:
                noised = ref_mcts.add_dirichlet_noise(p, eps=0.25, alpha=alpha)
            mask = rng.rand(A) < 0.7
            mask[0] = True
            masked = ref_mcts.set_illegal_action_probs_to_zero(mask, noised)
            masked32 = ref_mcts.set_illegal_action_probs_to_zero(mask, p)
            out[f'noise_{j}_p'] = p
            out[f'noise_{j}_noise'] = rec.noise
            out[f'noise_{j}_noised'] = noised
            out[f'noise_{j}_mask'] = mask.astype(np.uint8)
            out[f'noise_{j}_masked'] = masked
            out[f'noise_{j}_masked32'] = masked32
            assert noised.dtype == np.float64 and masked.dtype == np.float64 and masked32.dtype == np.float32
            j += 1
    out['noise_n'] = np.int32(j)
    # util.py: the reference KAT (tests/util_test.py:25-48) and sweeps
    x = torch.tensor([[3.7], [2.3]], dtype=torch.float32)
    out['twohot_kat_x'] = x.numpy()
    out['twohot_kat_out'] = ref_util.transform_to_2hot(x, -5, 5, 11).numpy()
    xs = torch.from_numpy(np.concatenate([np.linspace(-400, 400, 161), rng.uniform(-20, 20, 200), [0.0, 1e-6, -1e-6]]).astype(np.float32))
    out['xform_x'] = xs.numpy()
    out['xform_hyperbolic'] = ref_util.signed_hyperbolic(xs).numpy()
    out['xform_parabolic'] = ref_util.signed_parabolic(xs).numpy()
    for j, S in enumerate((31, 61, 601, 5)):
        lg = torch.from_numpy((rng.randn(16, S) * 3).astype(np.float32))
        out[f'logits_{j}_in'] = lg.numpy()
        out[f'logits_{j}_out'] = ref_util.logits_to_transformed_expected_value(lg, S).numpy()
        sc = torch.from_numpy(rng.uniform(-(S // 2) * 3.0, (S // 2) * 3.0, size=(4, 5)).astype(np.float32))
        out[f'cat_{j}_in'] = sc.numpy()
        out[f'cat_{j}_out'] = ref_util.scalar_to_categorical_probabilities(sc, S).numpy()
    h = torch.from_numpy(rng.randn(5, 64).astype(np.float32))
    out['norm_mlp_in'] = h.numpy()
    out['norm_mlp_out'] = ref_util.normalize_hidden_state(h).numpy()
    h = torch.from_numpy(rng.randn(2, 8, 3, 3).astype(np.float32))
    out['norm_conv_in'] = h.numpy()
    out['norm_conv_out'] = ref_util.normalize_hidden_state(h).numpy()
    # temperature schedules and config factory attributes (API surface, config.py:106-267)
    sched = []
    for fn in ('tictactoe', 'gomoku', 'classic', 'atari'):
        f = getattr(ref_config, f'{fn}_visit_softmax_temperature_fn')
        sched.append([f(es, ts) for es in (0, 5, 6, 29, 30, 100) for ts in (0, 29999, 30000, 60000, 499999, 500000, 1000000)])
    out['temperature_table'] = np.array(sched, np.float64)
    for fn in ('tictactoe', 'gomoku', 'classic', 'atari'):
        cfg = getattr(ref_config, f'make_{fn}_config')()
        fields = ['num_planes', 'num_res_blocks', 'value_support_size', 'reward_support_size', 'hidden_dim', 'num_simulations',
                  'discount', 'acc_seq_length', 'root_dirichlet_alpha', 'root_exploration_eps', 'pb_c_base', 'pb_c_init',
                  'num_training_steps', 'checkpoint_interval', 'min_replay_size', 'batch_size', 'unroll_steps', 'td_steps',
                  'weight_decay', 'momentum', 'max_grad_norm', 'lr_init', 'lr_decay_rate', 'train_delay']
        out[f'config_{fn}'] = np.array([float(getattr(cfg, f)) for f in fields], np.float64)
        out[f'config_{fn}_milestones'] = np.array(cfg.lr_milestones, np.float64)
        out[f'config_{fn}_flags'] = np.array([cfg.clip_grad, cfg.use_tensorboard, cfg.is_board_game, cfg.known_bounds is not None], np.int32)
    out['config_fields'] = np.array(fields)
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'pipe_cases.npz'), **out)
    print('pipe: done')


# --------------------------------------------------------------------------------------------
# G5: board-game env traces
# --------------------------------------------------------------------------------------------
def _play_trace(env, actions, prefix, out):
    obs0 = env.reset()
    obs, rew, done, mask, cur, winner = [obs0.astype(np.int8)], [], [], [env.actions_mask.copy()], [env.current_player], []
    for a in actions:
        o, r, d, _ = env.step(int(a))
        obs.append(o.astype(np.int8))
        rew.append(r)
        done.append(d)
        mask.append(env.actions_mask.copy())
        cur.append(env.current_player)
        winner.append(0 if env.winner is None else env.winner)
        if d:
            break
    n = len(rew)
    out[f'{prefix}_actions'] = np.array(actions[:n], np.int32)
    out[f'{prefix}_obs'] = np.stack(obs)
    out[f'{prefix}_reward'] = np.array(rew, np.float64)
    out[f'{prefix}_done'] = np.array(done, np.uint8)
    out[f'{prefix}_mask'] = np.stack(mask).astype(np.uint8)
    out[f'{prefix}_cur'] = np.array(cur, np.int32)
    out[f'{prefix}_winner'] = np.array(winner, np.int32)


def gen_env():
    rng = np.random.RandomState(31)
    out = {}
    j = 0
    # the 8 TicTacToe win lines for each colour (tests/games/tictactoe_test.py:25-34): winner plays the line,
    # the other side plays the remaining cells in index order.
    lines = [(0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 3, 6), (1, 4, 7), (2, 5, 8), (0, 4, 8), (2, 4, 6)]
    for line in lines:
        for colour in (1, 2):
            others = [c for c in range(9) if c not in line]
            # pick filler moves that do not themselves complete a line
            seq = []
            li, oi = 0, 0
            filler = [c for c in others]
            rng.shuffle(filler)
            mover_is_winner = (colour == 1)
            while li < 3:
                if mover_is_winner:
                    seq.append(line[li])
                    li += 1
                else:
                    seq.append(filler[oi])
                    oi += 1
                mover_is_winner = not mover_is_winner
            env = TicTacToeEnv()
            _play_trace(env, seq, f'ttt_{j}', out)
            j += 1
    # random legal games incl. resign (action 9, tests/games/boardgame_test.py:42-55) and draws
    for g in range(24):
        env = TicTacToeEnv()
        env.reset()
        seq = []
        sim = TicTacToeEnv()
        sim.reset()
        while not sim.is_game_over:
            legal = np.where(sim.actions_mask)[0]
            if g % 6 != 0:
                legal = legal[legal != 9]
            a = int(rng.choice(legal))
            seq.append(a)
            sim.step(a)
        _play_trace(env, seq, f'ttt_{j}', out)
        j += 1
    out['ttt_n'] = np.int32(j)
    # Gomoku: 9x9 (launcher default) and 15x15; num_to_win 5, stack 4 (gomoku/run_training.py)
    j = 0
    for board_size in (9, 15):
        for g in range(5):
            env = GomokuEnv(board_size=board_size, stack_history=4)
            env.reset()
            sim = GomokuEnv(board_size=board_size, stack_history=4)
            sim.reset()
            seq = []
            if g < 3:
                # scripted five-in-row for black / white along row, column, diagonal
                base = board_size * 2 + 2
                step = [1, board_size, board_size + 1][g]
                win = [base + i * step for i in range(5)]
                other = [board_size * (board_size - 1) + i * 2 for i in range(5)]
                black_wins = (g % 2 == 0)
                wi = oi = 0
                mover_black = True
                while wi < 5:
                    if mover_black == black_wins:
                        seq.append(win[wi])
                        wi += 1
                    else:
                        seq.append(other[oi])
                        oi += 1
                    mover_black = not mover_black
            else:
                while not sim.is_game_over and len(seq) < 60:
                    legal = np.where(sim.actions_mask)[0]
                    legal = legal[legal != board_size * board_size] if g == 3 else legal
                    a = int(rng.choice(legal))
                    seq.append(a)
                    sim.step(a)
            _play_trace(env, seq, f'gomoku_{j}', out)
            out[f'gomoku_{j}_board'] = np.int32(board_size)
            j += 1
    out['gomoku_n'] = np.int32(j)
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'env_cases.npz'), **out)
    print('env: done')


# --------------------------------------------------------------------------------------------
# G6: one full reference run_self_play episode (pipeline.py:41-167) on TicTacToe with the
# seeded MLP; every search's draws are recorded so the episode can be replayed exactly.
# --------------------------------------------------------------------------------------------
def gen_play():
    out = {}
    for ep, seed in enumerate((4242, 4243, 4244)):
        case = MLP_CASES[2]
        net = build_mlp(case)
        cfg = ref_config.make_tictactoe_config(use_tensorboard=False)
        env = TicTacToeEnv()
        records = []
        searches = []

        class Q:
            def put(self, item):
                records.append(item)
                stop.flag = True

        class Stop:
            flag = False

            def is_set(self):
                return self.flag

        stop = Stop()
        counter = types.SimpleNamespace(value=0)
        orig_search = ref_pipeline.uct_search

        def logged_search(**kw):
            with DrawRecorder() as rec:
                res = orig_search(**kw)
            d = rec.as_dict(10, max_ties=4 * cfg.num_simulations + 8)
            d.update(
                obs=np.asarray(kw['state'], np.int8),
                mask=np.asarray(kw['actions_mask']).astype(np.uint8).copy(),
                cur=np.int32(kw['current_player']),
                opp=np.int32(kw['opponent_player']),
                T=np.float64(kw['temperature']),
                action=np.int32(res[0]),
                pi=np.asarray(res[1], np.float64),
                root=np.float64(res[2]),
            )
            searches.append(d)
            return res

        ref_pipeline.uct_search = logged_search
        ref_pipeline.handle_exit_signal = lambda: None
        np.random.seed(seed)
        try:
            ref_pipeline.run_self_play(cfg, 0, net, torch.device('cpu'), env, Q(), counter, stop, None)
        finally:
            ref_pipeline.uct_search = orig_search
        n = len(searches)
        for k in searches[0].keys():
            out[f'ep{ep}_search_{k}'] = np.stack([np.asarray(s[k]) for s in searches])
        out[f'ep{ep}_n_moves'] = np.int32(n)
        out[f'ep{ep}_tr_state'] = np.stack([t.state for t, _ in records])
        out[f'ep{ep}_tr_action'] = np.stack([t.action for t, _ in records])
        out[f'ep{ep}_tr_reward'] = np.stack([t.reward for t, _ in records])
        out[f'ep{ep}_tr_value'] = np.stack([t.value for t, _ in records])
        out[f'ep{ep}_tr_pi'] = np.stack([t.pi_prob for t, _ in records])
        out[f'ep{ep}_tr_priority'] = np.array([p for _, p in records], np.float64)
    out['n_episodes'] = np.int32(3)
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'selfplay_cases.npz'), **out)
    print('play: done')


# --------------------------------------------------------------------------------------------
# L: learner side: the reference's PrioritizedReplay and calc_loss / optimizer steps run here
# --------------------------------------------------------------------------------------------
LEARN_CASES = [
    # name, kind, case, batch
    ('mlp_cat', 'mlp', 'tiny', 6),             # categorical value / reward heads
    ('mlp_mse', 'mlp', 'tiny_mse', 5),         # MSE heads
    ('conv_board3', 'conv', 'board3', 4),      # BatchNorm in train mode
    ('conv_atari_s', 'conv', 'atari_s', 3),    # MuZeroAtariNet: strided 96 x 96 representation, average pools, categorical conv heads (round 5)
]


def _learn_batch(rng, net, ishape, A, B, K=5):
    state = rng.uniform(-1.0, 1.0, size=(B,) + tuple(ishape)).astype(np.float32)
    action = rng.randint(0, A, size=(B, K)).astype(np.int8)
    pi = rng.dirichlet(np.ones(A), size=(B, K)).astype(np.float32)
    value = rng.uniform(-3.0, 3.0, size=(B, K)).astype(np.float32)
    reward = rng.uniform(-1.0, 1.0, size=(B, K)).astype(np.float32)
    weights = rng.uniform(0.2, 1.0, size=B).astype(np.float32)
    return ref_replay.Transition(state=state, action=action, pi_prob=pi, value=value, reward=reward), weights


def gen_learn():
    import muzero.replay as _rr
    global ref_replay
    ref_replay = _rr
    out = {}
    rng = np.random.RandomState(909)
    # ---- replay: uniform (exponent 0: RandomState.uniform) and prioritized (global np.random.choice) sampling
    for j, (cap, n_add, pexp, isexp) in enumerate([(16, 11, 0.0, 0.0), (8, 21, 0.0, 0.0), (16, 16, 0.6, 0.4), (8, 13, 1.0, 1.0)]):
        rp = ref_replay.PrioritizedReplay(cap, pexp, isexp, np.random.RandomState(5 + j))
        items, prios = [], []
        for i in range(n_add):
            tr = ref_replay.Transition(state=rng.uniform(-1, 1, size=(3, 4)).astype(np.float32), action=rng.randint(0, 4, size=5).astype(np.int8),
                                       pi_prob=rng.dirichlet(np.ones(4), size=5).astype(np.float32), value=rng.uniform(-1, 1, size=5).astype(np.float32),
                                       reward=rng.uniform(-1, 1, size=5).astype(np.float32))
            pr = float(rng.uniform(0.01, 2.0))
            rp.add(tr, pr)
            items.append(tr)
            prios.append(pr)
        np.random.seed(100 + j)
        batch, idx, w = rp.sample(6)
        rp.update_priorities(idx[:3], [0.5, 1.5, 2.5])
        np.random.seed(200 + j)
        batch2, idx2, w2 = rp.sample(4)
        pre = f'replay_{j}'
        out[f'{pre}_cfg'] = np.array([cap, n_add, pexp, isexp], np.float64)
        for f in ref_replay.Transition._fields:
            out[f'{pre}_items_{f}'] = np.stack([getattr(t, f) for t in items])
            out[f'{pre}_s1_{f}'] = getattr(batch, f)
            out[f'{pre}_s2_{f}'] = getattr(batch2, f)
        out[f'{pre}_prios'] = np.array(prios, np.float64)
        out[f'{pre}_s1_idx'], out[f'{pre}_s1_w'] = np.asarray(idx, np.int64), np.asarray(w, np.float32)
        out[f'{pre}_s2_idx'], out[f'{pre}_s2_w'] = np.asarray(idx2, np.int64), np.asarray(w2, np.float32)
        out[f'{pre}_size'] = np.int64(rp.size)
    out['replay_n'] = np.int32(4)
    # ---- calc_loss + 3 Adam / MultiStepLR steps (pipeline.py:170-286,541-629)
    for name, kind, cname, B in LEARN_CASES:
        case = helpers.mlp_case(cname) if kind == 'mlp' else helpers.conv_case(cname)
        net = build_mlp(case) if kind == 'mlp' else build_conv(case)
        ishape, A = (case[1], case[2]) if kind == 'mlp' else (case[2], case[3])
        net.train()
        tr, weights = _learn_batch(rng, net, ishape, A, B)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[2], gamma=0.1)
        pre = f'learn_{name}'
        for f in ref_replay.Transition._fields:
            out[f'{pre}_{f}'] = getattr(tr, f)
        out[f'{pre}_weights'] = weights
        losses = []
        for step in range(3):
            opt.zero_grad()
            loss, prio = ref_pipeline.calc_loss(net, torch.device('cpu'), tr, torch.from_numpy(weights))
            loss.backward()
            if step == 0:
                out[f'{pre}_prio'] = np.asarray(prio, np.float32)
                for pn, pp in net.named_parameters():
                    out[f'{pre}_grad_{pn}'] = pp.grad.detach().numpy().copy()
            if step == 1:
                torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0)
            opt.step()
            sched.step()
            losses.append(float(loss.detach()))
        out[f'{pre}_losses'] = np.array(losses, np.float64)
        for pn, pp in net.state_dict().items():
            out[f'{pre}_final_{pn}'] = pp.detach().numpy().copy()
    # ---- target projection (util.py:96-116): scalar -> 2-hot categorical
    xs = torch.tensor([[-12.3, -1.0, -0.2, 0.0], [0.3, 1.0, 7.7, 300.0]])
    for S in (31, 61, 601):
        out[f'proj_{S}'] = ref_util.scalar_to_categorical_probabilities(xs, S).numpy()
    out['proj_x'] = xs.numpy()
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'learn_cases.npz'), **out)
    print('learn: done,', len(out), 'arrays')


# --------------------------------------------------------------------------------------------
# G7: classic-control observation path and the n-step / mid-episode-flush branch of run_self_play
# --------------------------------------------------------------------------------------------
def gen_classic():
    import gym  # the placeholder module of _refshim: Env / Wrapper / spaces only
    import muzero.gym_env as ref_gym_env

    class ScriptedEnv(gym.Env):
        """Stands in for gym's CartPole (absent from this image): a fixed table of float32 observations, rewards and
        episode lengths.  The wrappers under test only see reset() / step() / the two spaces."""

        def __init__(self, obs, rewards):
            self.obs, self.rewards = obs, rewards  # lists per episode: [T+1, D] float32, [T] float64
            self.observation_space = gym.spaces.Box(low=np.full(obs[0].shape[1], -4.0, np.float32), high=np.full(obs[0].shape[1], 4.0, np.float32),
                                                    shape=(obs[0].shape[1],), dtype=np.float32)
            self.action_space = gym.spaces.Discrete(2)
            self.ep, self.t = -1, 0
            self.actions = []

        def reset(self, **kwargs):
            self.ep += 1
            self.t = 0
            return self.obs[self.ep % len(self.obs)][0]

        def step(self, action):
            e = self.ep % len(self.obs)
            self.actions.append(int(action))
            self.t += 1
            done = self.t == len(self.rewards[e])
            return self.obs[e][self.t], float(self.rewards[e][self.t - 1]), done, {}

    out = {}
    rng = np.random.RandomState(777)

    def table(lengths, D=4):
        return ([rng.uniform(-2.0, 2.0, size=(T + 1, D)).astype(np.float32) for T in lengths],
                [np.round(rng.uniform(0.0, 2.0, size=T), 3) for T in lengths])

    # ---- (a) StackFrameAndAction(stack, is_obs_image=False) + PlayerIdAndActionMaskWrapper: reset / step stacks ----
    for j, (stack, lengths) in enumerate([(4, (6, 3, 5)), (1, (2, 2)), (3, (9,))]):
        obs, rew = table(lengths)
        base = ScriptedEnv(obs, rew)
        env = ref_gym_env.PlayerIdAndActionMaskWrapper(ref_gym_env.StackFrameAndAction(base, stack, False))  # __init__ consumes one reset()
        pre = f'stack_{j}'
        out[f'{pre}_cfg'] = np.array([stack, len(lengths)], np.int32)
        out[f'{pre}_obs_space_shape'] = np.array(env.observation_space.shape, np.int32)
        out[f'{pre}_mask'] = np.asarray(env.actions_mask).astype(np.uint8)
        out[f'{pre}_players'] = np.array([env.current_player, env.opponent_player], np.int32)
        stacks, acts, rws, dns, eps = [], [], [], [], []
        for e in range(len(lengths)):
            # episode e of the wrapped env plays table row (e + 1) % n: the constructor used up row 0's reset
            o = env.reset()
            stacks.append(np.asarray(o, np.float32).copy()); acts.append(-1); rws.append(0.0); dns.append(0); eps.append(base.ep % len(obs))
            done = False
            while not done:
                a = int(rng.randint(0, 2))
                o, r, done, _ = env.step(a)
                stacks.append(np.asarray(o, np.float32).copy()); acts.append(a); rws.append(r); dns.append(int(done)); eps.append(base.ep % len(obs))
        out[f'{pre}_stacks'] = np.stack(stacks)
        out[f'{pre}_actions'] = np.array(acts, np.int32)   # -1: the row is a reset() output
        out[f'{pre}_rewards'] = np.array(rws, np.float64)
        out[f'{pre}_dones'] = np.array(dns, np.uint8)
        out[f'{pre}_table_row'] = np.array(eps, np.int32)
        for e in range(len(lengths)):
            out[f'{pre}_base_obs_{e}'] = obs[e]
            out[f'{pre}_base_rew_{e}'] = rew[e]
    out['stack_n'] = np.int32(3)

    # ---- (b) reference run_self_play on the wrapped scripted env: n-step targets + mid-episode flush (pipeline.py:118-142) ----
    for j, (acc, td, lengths, seed) in enumerate([(4, 3, (30, 12, 7), 5150), (6, 10, (17, 23), 5151), (3, 2, (10, 10, 10), 5152)]):
        net = ref_network.MuZeroMLPNet((4, 5), 2, 32, 7, 7, 16)
        net.load_state_dict(seeded_state_dict(net, 31 + j))
        net.eval()
        cfg = ref_config.MuZeroConfig(discount=0.997, dirichlet_alpha=0.25, num_simulations=6, batch_size=8, td_steps=td, lr_init=0.01,
                                      lr_milestones=[10], visit_softmax_temperature_fn=lambda a, b: 1.0, value_support_size=7,
                                      reward_support_size=7, acc_seq_length=acc, use_tensorboard=False, is_board_game=False)
        obs, rew = table(lengths)
        base = ScriptedEnv(obs, rew)
        env = ref_gym_env.PlayerIdAndActionMaskWrapper(ref_gym_env.StackFrameAndAction(base, 4, False))
        n_episodes = len(lengths)
        records, steps = [], []

        class Stop:
            def is_set(self):
                # stop once `n_episodes` episodes have been played AND flushed (the check runs at the top of both loops)
                return base.ep >= n_episodes and base.t == len(rew[base.ep % len(obs)])

        class Q:
            def put(self, item):
                records.append((item, len(steps)))

        counter = types.SimpleNamespace(value=0)
        orig_search = ref_pipeline.uct_search

        def logged_search(**kw):
            res = orig_search(**kw)
            steps.append(dict(obs=np.asarray(kw['state'], np.float32).copy(), action=np.int32(res[0]), pi=np.asarray(res[1], np.float64),
                              root=np.float64(res[2]), player=np.int32(kw['current_player'])))
            return res

        ref_pipeline.uct_search = logged_search
        ref_pipeline.handle_exit_signal = lambda: None
        np.random.seed(seed)
        try:
            ref_pipeline.run_self_play(cfg, 0, net, torch.device('cpu'), env, Q(), counter, Stop(), None)
        finally:
            ref_pipeline.uct_search = orig_search
        pre = f'sp_{j}'
        n = len(steps)
        # per-step env outputs in play order: the scripted table replayed with the recorded actions
        rws, dns = [], []
        e, t = 1 % len(obs), 0  # constructor consumed row 0's reset; the first played episode is table row 1 % n
        for k in range(n):
            rws.append(rew[e][t]); t += 1
            d = t == len(rew[e]); dns.append(int(d))
            if d:
                e, t = (e + 1) % len(obs), 0
        out[f'{pre}_cfg'] = np.array([acc, td, cfg.unroll_steps, n], np.int32)
        out[f'{pre}_discount'] = np.float64(cfg.discount)
        for k in steps[0].keys():
            out[f'{pre}_step_{k}'] = np.stack([np.asarray(s_[k]) for s_ in steps])
        out[f'{pre}_step_reward'] = np.array(rws, np.float64)
        out[f'{pre}_step_done'] = np.array(dns, np.uint8)
        out[f'{pre}_tr_state'] = np.stack([t_.state for (t_, _), _ in records])
        out[f'{pre}_tr_action'] = np.stack([t_.action for (t_, _), _ in records])
        out[f'{pre}_tr_reward'] = np.stack([t_.reward for (t_, _), _ in records])
        out[f'{pre}_tr_value'] = np.stack([t_.value for (t_, _), _ in records])
        out[f'{pre}_tr_pi'] = np.stack([t_.pi_prob for (t_, _), _ in records])
        out[f'{pre}_tr_priority'] = np.array([p_ for (_, p_), _ in records], np.float64)
        out[f'{pre}_tr_emitted_after_step'] = np.array([k for _, k in records], np.int32)  # env steps played when the item was put
    out['sp_n'] = np.int32(3)
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'classic_cases.npz'), **out)
    print('classic: done,', len(out), 'arrays')



# --------------------------------------------------------------------------------------------
# G3': the numpy-1.21 form of child_U for searches without root noise.  In mcts.py:189-197 `child.prior` is an
# np.float32 scalar there and the other factor a Python float: numpy >= 2 (NEP 50) multiplies in float32, numpy 1.21.6 -- the
# reference's pinned version -- promotes scalar x scalar to float64 and np.array(..., dtype=np.float32) rounds once.  numpy 1.21 cannot
# be installed here, but its result can be produced exactly: with the child priors stored as FLOAT64 scalars (the float32 values
# widened, which is exact) numpy 2 evaluates the very same float64 product.  Nothing else in uct_search depends on the scalar type.
# --------------------------------------------------------------------------------------------
class _Numpy121Promotion:
    def __enter__(self):
        self._orig = ref_mcts.Node.expand
        orig = self._orig

        def expand(node, prior, player_id, hidden_state, reward):
            return orig(node, np.asarray(prior).astype(np.float64), player_id, hidden_state, reward)

        ref_mcts.Node.expand = expand
        return self

    def __exit__(self, *exc):
        ref_mcts.Node.expand = self._orig


def _legacy_case(net, cfg, obs, mask, players, prefix, out, A, stats):
    """One deterministic search in both promotion forms on the same root; the legacy one is stored, the numpy-2 one only compared."""
    tmp = {}
    _search_case(net, cfg, obs, mask, players, 1.0, True, 7000, 'n2', tmp, A)
    with _Numpy121Promotion():
        _search_case(net, cfg, obs, mask, players, 1.0, True, 7000, prefix, out, A)
    same = np.array_equal(tmp['n2_visits'], out[f'{prefix}_visits']) and tmp['n2_out_action'] == out[f'{prefix}_out_action']
    out[f'{prefix}_same_as_numpy2'] = np.int32(same)
    stats.append(same)


def gen_legacy():
    out, stats = {}, []
    rng = np.random.RandomState(777)
    # seeded random-weight nets (rebuilt from their seeds wherever the tests run)
    for gname, case, kind, cfgargs, n_roots in [('cartpole', MLP_CASES[0], 'mlp', (0.997, 0.25, 50, False, None, 31, 31), 12),
                                                ('tictactoe', MLP_CASES[2], 'mlp', (1.0, 0.25, 25, True, (-1, 1), 1, 1), 12),
                                                ('board3', CONV_CASES[0], 'conv', (1.0, 0.25, 20, True, (-1, 1), 1, 1), 6)]:
        net = build_mlp(case) if kind == 'mlp' else build_conv(case)
        cfg = make_config(*cfgargs)
        for k, v in cfg_arrays(cfg).items():
            out[f'{gname}_{k}'] = v
        A = case[2] if kind == 'mlp' else case[3]
        for j in range(n_roots):
            if gname == 'cartpole':
                obs = rng.uniform(-0.5, 0.5, size=(4, 5)).astype(np.float32)
                obs[:, 4] = (rng.randint(0, 2, size=4) + 1) / 2.0
                mask, players = np.ones(2, bool), (1, 1)
            else:
                env = TicTacToeEnv()
                obs = env.reset()
                for _ in range(rng.randint(0, 5)):
                    obs, _, done, _ = env.step(int(rng.choice(np.where(env.actions_mask[:9])[0])))
                    if done:
                        break
                if env.is_game_over:
                    env = TicTacToeEnv()
                    obs = env.reset()
                obs = obs.astype(np.int8) if gname == 'tictactoe' else obs.astype(np.float32)
                mask, players = env.actions_mask.copy(), (env.current_player, env.opponent_player)
            _legacy_case(net, cfg, obs, mask, players, f'{gname}_{j}', out, A, stats)
        out[f'{gname}_n'] = np.int32(n_roots)
    # the shipped trained checkpoints (where near-ties live); replayed by the container-only test
    for name, fname, ishape, A, P, sup, (disc, sims, board, bounds) in CKPT_CASES:
        net = ref_network.MuZeroMLPNet(ishape, A, P, sup, sup, 64)
        net.load_state_dict(torch.load(os.path.join(CKPT_DIR, fname), map_location='cpu', weights_only=False)['network'])
        net.eval()
        cfg = make_config(disc, 0.25, sims, board, bounds, sup, sup)
        for k, v in cfg_arrays(cfg).items():
            out[f'ckpt_{name}_{k}'] = v
        for j in range(16):
            if board:
                env = TicTacToeEnv()
                obs = env.reset()
                for _ in range(rng.randint(0, 6)):
                    obs, _, done, _ = env.step(int(rng.choice(np.where(env.actions_mask[:9])[0])))
                    if done:
                        break
                if env.is_game_over:
                    env = TicTacToeEnv()
                    obs = env.reset()
                obs, mask, players = obs.astype(np.int8), env.actions_mask.copy(), (env.current_player, env.opponent_player)
            else:
                D = ishape[1] - 1
                obs = rng.uniform(-0.2 if name == 'cartpole' else -1.0, 0.2 if name == 'cartpole' else 1.0, size=ishape).astype(np.float32)
                obs[:, D] = (rng.randint(0, A, size=ishape[0]) + 1) / float(A)
                mask, players = np.ones(A, bool), (1, 1)
            _legacy_case(net, cfg, obs, mask, players, f'ckpt_{name}_{j}', out, A, stats)
        out[f'ckpt_{name}_n'] = np.int32(16)
    # scripted trees (G1 format, prefix lt<i>_) where the two forms provably DIFFER: priors in pairs one to three float32 ulps apart, so that
    # child_U products fall on either side of a float32 rounding boundary depending on where the product is rounded.  The first 8 searches
    # whose per-simulation (parent, action) trace differs between the forms are kept, plus 2 that agree.
    hunt = np.random.RandomState(5)
    kept_diff = kept_same = 0
    ci = 0
    for trial in range(2000):
        if kept_diff >= 8 and kept_same >= 2:
            break
        A = int(hunt.choice([4, 10, 30]))
        S = 40
        base = hunt.randn(A // 2 + 1).astype(np.float32)
        logits = np.repeat(base, 2)[:A].copy()
        logits[1::2] += (hunt.randint(-3, 4, size=logits[1::2].shape) * 1e-7).astype(np.float32)
        pi0 = torch.softmax(torch.from_numpy(logits), dim=0).numpy()
        vk = trial % 3
        values = (np.zeros(S) if vk == 0 else hunt.uniform(-1, 1, S) * (1e-3 if vk == 1 else 1.0)).astype(np.float32)
        rewards = np.zeros(S, np.float32)
        board = bool(trial % 2)
        cfg = make_config(1.0 if board else 0.997, 0.0, S, board, (-1, 1) if board else None)
        players = (1, 2) if board else (1, 1)
        runs = []
        for legacy in (False, True):
            net = ScriptedNet(pi0, values, rewards)
            np.random.seed(9000 + trial)
            with DrawRecorder() as rec:
                if legacy:
                    with _Numpy121Promotion():
                        res = ref_mcts.uct_search(state=np.zeros((1,), np.float32), network=net, device=torch.device('cpu'), config=cfg, temperature=1.0,
                                                  actions_mask=np.ones(A, bool), current_player=players[0], opponent_player=players[1], deterministic=True)
                else:
                    res = ref_mcts.uct_search(state=np.zeros((1,), np.float32), network=net, device=torch.device('cpu'), config=cfg, temperature=1.0,
                                              actions_mask=np.ones(A, bool), current_player=players[0], opponent_player=players[1], deterministic=True)
            runs.append((net, res, rec))
        differs = runs[0][0].trace_parent != runs[1][0].trace_parent or runs[0][0].trace_action != runs[1][0].trace_action
        if (differs and kept_diff >= 8) or (not differs and kept_same >= 2):
            continue
        kept_diff += int(differs)
        kept_same += int(not differs)
        net, (action, pi, root_value), rec = runs[1]
        d = dict(A=np.int32(A), pi0=pi0, values=values, rewards=rewards, mask=np.ones(A, np.uint8), temperature=np.float64(1.0), deterministic=np.int32(1),
                 cur_player=np.int32(players[0]), opp_player=np.int32(players[1]), trace_parent=np.array(net.trace_parent, np.int32),
                 trace_action=np.array(net.trace_action, np.int32), out_action=np.int32(action), out_pi=np.asarray(pi, np.float64),
                 out_root_value=np.float64(root_value), differs_from_numpy2=np.int32(differs), seed=np.int64(9000 + trial),
                 numpy2_trace_parent=np.array(runs[0][0].trace_parent, np.int32), numpy2_trace_action=np.array(runs[0][0].trace_action, np.int32))
        d.update(cfg_arrays(cfg))
        d.update(rec.as_dict(A, max_ties=4 * S + 8))
        for k, v in d.items():
            out[f'lt{ci}_{k}'] = v
        ci += 1
    out['lt_n'] = np.int32(ci)
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'legacy_cases.npz'), **out)
    print('legacy: done,', len(out), 'arrays;', len(stats) - int(np.sum(stats)), 'of', len(stats), 'network searches and', kept_diff, 'of', ci,
          'scripted trees differ from the numpy-2 form')


GROUPS = dict(legacy=gen_legacy, classic=gen_classic, tree=gen_tree, nets=gen_nets, search=gen_search, ckpt=gen_ckpt, pipe=gen_pipe, env=gen_env, play=gen_play, learn=gen_learn)

if __name__ == '__main__':
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    which = sys.argv[1:] or list(GROUPS)
    for g in which:
        GROUPS[g]()
