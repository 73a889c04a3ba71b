"""G7: the classic-control observation path and the n-step / mid-episode-flush branch of run_self_play, against vectors
recorded from the reference's own StackFrameAndAction + PlayerIdAndActionMaskWrapper (gym_env.py:271-365) and
run_self_play (pipeline.py:41-167) on a scripted base env (oracle/gen_golden.py classic)."""
import ctypes as C
import types

import numpy as np
import pytest

from helpers import load_golden

G = load_golden('classic_cases.npz')


def _episodes(pre):
    """Split the recorded reset()/step() outputs of stack case `pre` into (table row, stacks, actions, rewards, dones)."""
    acts = G[f'{pre}_actions']
    starts = list(np.nonzero(acts == -1)[0]) + [len(acts)]
    for a, b in zip(starts[:-1], starts[1:]):
        yield int(G[f'{pre}_table_row'][a]), G[f'{pre}_stacks'][a:b], acts[a + 1:b], G[f'{pre}_rewards'][a + 1:b], G[f'{pre}_dones'][a + 1:b]


@pytest.mark.parametrize('j', range(int(G['stack_n'])))
def test_oracle_stacker_matches_reference_wrapper(oracle, j):
    """mzo_stack_reset / mzo_stack_push (the oracle's StackFrameAndAction) on the reference's scripted episodes: exact."""
    import oracle as orc

    pre = f'stack_{j}'
    stack = int(G[f'{pre}_cfg'][0])
    lib = orc.lib()
    for row, stacks, acts, _, _ in _episodes(pre):
        base = G[f'{pre}_base_obs_{row}']
        D = base.shape[1]
        st = np.zeros((stack, D + 1), np.float32)
        lib.mzo_stack_reset(st.ctypes.data_as(C.c_void_p), stack, D, np.ascontiguousarray(base[0]).ctypes.data_as(C.c_void_p), 2)
        np.testing.assert_array_equal(st, stacks[0])
        for t, a in enumerate(acts):
            lib.mzo_stack_push(st.ctypes.data_as(C.c_void_p), stack, D, np.ascontiguousarray(base[t + 1]).ctypes.data_as(C.c_void_p), int(a), 2)
            np.testing.assert_array_equal(st, stacks[t + 1])


class _Scripted:
    """The generator's scripted base env, rebuilt from the fixture's tables."""
    num_actions = 2

    def __init__(self, pre, n_rows):
        self.obs = [G[f'{pre}_base_obs_{e}'] for e in range(n_rows)]
        self.rew = [G[f'{pre}_base_rew_{e}'] for e in range(n_rows)]
        self.observation_shape = (self.obs[0].shape[1],)
        self.ep, self.t = -1, 0

    def reset(self, **kw):
        self.ep += 1
        self.t = 0
        return self.obs[self.ep % len(self.obs)][0]

    def step(self, a):
        e = self.ep % len(self.obs)
        self.t += 1
        return self.obs[e][self.t], float(self.rew[e][self.t - 1]), self.t == len(self.rew[e]), {}


@pytest.mark.parametrize('j', range(int(G['stack_n'])))
def test_host_wrappers_match_reference_wrappers(j):
    """muzero_amd.games.StackFrameAndAction + PlayerIdAndActionMaskWrapper driven exactly like the reference's wrappers
    were when the fixture was recorded (the reference's constructor consumes one reset(): so does this test)."""
    from muzero_amd import games

    pre = f'stack_{j}'
    stack, n_rows = (int(v) for v in G[f'{pre}_cfg'])
    base = _Scripted(pre, n_rows)
    env = games.PlayerIdAndActionMaskWrapper(games.StackFrameAndAction(base, stack, False))
    env.reset()  # what StackFrameAndAction.__init__ does in the reference (gym_env.py:304)
    assert tuple(env.observation_shape) == tuple(G[f'{pre}_obs_space_shape'])
    np.testing.assert_array_equal(np.asarray(env.actions_mask, np.uint8), G[f'{pre}_mask'])
    assert [env.current_player, env.opponent_player] == list(G[f'{pre}_players'])
    for row, stacks, acts, rews, dones in _episodes(pre):
        o = env.reset()
        assert o.dtype == np.float32
        np.testing.assert_array_equal(o, stacks[0])
        for t, a in enumerate(acts):
            o, r, d, _ = env.step(int(a))
            np.testing.assert_array_equal(o, stacks[t + 1])
            assert r == rews[t] and bool(d) == bool(dones[t])


def _cfg(j):
    acc, td, unroll, n = (int(v) for v in G[f'sp_{j}_cfg'])
    return types.SimpleNamespace(is_board_game=False, acc_seq_length=acc, td_steps=td, unroll_steps=unroll, discount=float(G[f'sp_{j}_discount'])), n


@pytest.mark.parametrize('j', range(int(G['sp_n'])))
@pytest.mark.parametrize('chunk', [1, 5, 1000])
def test_episode_assembler_reproduces_reference_self_play_items(j, chunk):
    """EpisodeAssembler fed the reference's own per-step tuples (in record-ring sized chunks): every (Transition, priority)
    the reference put on its queue, in the same order and after the same env step -- including the mid-episode flush
    (pipeline.py:118-142) and the step where flush and episode end coincide."""
    from muzero_amd.pipeline import EpisodeAssembler

    cfg, n = _cfg(j)
    pre = f'sp_{j}'
    asm = EpisodeAssembler(cfg, 1, (4, 5))
    items, after = [], []
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        rec = dict(obs=G[f'{pre}_step_obs'][lo:hi].reshape(hi - lo, 1, 20), action=G[f'{pre}_step_action'][lo:hi, None],
                   reward=G[f'{pre}_step_reward'][lo:hi, None], pi=G[f'{pre}_step_pi'][lo:hi, None], root_value=G[f'{pre}_step_root'][lo:hi, None],
                   player=G[f'{pre}_step_player'][lo:hi, None], done=G[f'{pre}_step_done'][lo:hi, None])
        # the assembler yields lazily inside its move loop: count the moves consumed when each item comes out
        consumed = {'m': lo}
        orig = rec['action']

        class Probe:
            shape = orig.shape

            def __getitem__(self, idx):
                consumed['m'] = lo + idx[0] + 1
                return orig[idx]

        rec['action'] = Probe()
        for it in asm.feed(rec):
            items.append(it)
            after.append(consumed['m'])
    assert len(items) == len(G[f'{pre}_tr_priority'])
    np.testing.assert_array_equal(np.array(after), G[f'{pre}_tr_emitted_after_step'])
    for i, (tr, prio) in enumerate(items):
        np.testing.assert_array_equal(tr.state, G[f'{pre}_tr_state'][i])
        np.testing.assert_array_equal(tr.action, G[f'{pre}_tr_action'][i])
        assert tr.action.dtype == np.int8
        np.testing.assert_array_equal(tr.reward, G[f'{pre}_tr_reward'][i])
        np.testing.assert_array_equal(tr.value, G[f'{pre}_tr_value'][i])
        np.testing.assert_array_equal(tr.pi_prob, G[f'{pre}_tr_pi'][i])
        assert prio == G[f'{pre}_tr_priority'][i]


@pytest.mark.parametrize('j', range(int(G['sp_n'])))
def test_oracle_target_builders_on_reference_episodes(oracle, j):
    """The oracle's n-step targets / unroll windows (C restatement of pipeline.py:632-767) on the same episodes: the items of
    every episode's final flush."""
    cfg, n = _cfg(j)
    pre = f'sp_{j}'
    done = G[f'{pre}_step_done']
    ends = list(np.nonzero(done)[0] + 1)
    emitted = G[f'{pre}_tr_emitted_after_step']
    lo = 0
    for hi in ends:
        T = hi - lo
        rew = [float(x) for x in G[f'{pre}_step_reward'][lo:hi]]
        roots = [float(x) for x in G[f'{pre}_step_root'][lo:hi]]
        z = oracle.n_step_target(rew, roots, cfg.td_steps, cfg.discount)
        # the items emitted at the episode's last step cover the tail of the episode; their step-0 values are the targets
        idx = np.nonzero(emitted == hi)[0]
        tail = T - len(idx) if len(idx) <= T else 0
        got = G[f'{pre}_tr_value'][idx][-(T - tail):, 0] if tail else G[f'{pre}_tr_value'][idx][:, 0]
        np.testing.assert_array_equal(np.array(z[tail:], np.float32)[-len(got):], got)
        lo = hi
