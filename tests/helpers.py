"""Shared test helpers: seeded weights and the network case tables used by oracle/gen_golden.py (to record
reference outputs) and by the tests (to rebuild the very same weights on any machine from a seed)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, 'tests', 'golden')
if REPO not in sys.path:
    sys.path.insert(0, REPO)

MLP_CASES = [
    # name, input_shape, A, planes, value_support, reward_support, hidden, seed
    ('cartpole', (4, 5), 2, 512, 31, 31, 64, 11),
    ('lunar', (4, 9), 4, 512, 31, 31, 64, 12),
    ('tictactoe', (9, 3, 3), 10, 256, 1, 1, 64, 13),
    ('tiny', (3, 4), 3, 32, 7, 5, 16, 14),
    ('tiny_mse', (2, 2, 2), 5, 24, 1, 1, 8, 15),
    ('odd', (7,), 6, 40, 31, 1, 20, 16),
]

CONV_CASES = [
    # name, kind, input_shape, A, blocks, planes, value_support, reward_support, seed
    ('board3', 'board', (9, 3, 3), 10, 2, 16, 1, 1, 21),
    ('board5', 'board', (5, 5, 5), 26, 1, 8, 1, 1, 22),
    ('board9', 'board', (9, 9, 9), 82, 1, 8, 1, 1, 23),
    ('atari_s', 'atari', (4, 96, 96), 6, 1, 8, 11, 11, 24),
    ('atari_m', 'atari', (8, 96, 96), 4, 2, 16, 61, 61, 25),
]


def seeded_state_dict(module, seed):
    """Deterministic weights for `module` from numpy's legacy RandomState (stream frozen by NEP 19): one
    standard_normal draw per tensor in state_dict order, scaled by tensor role.  Works on the reference's
    modules and on muzero_amd.network's (identical key names / order / shapes)."""
    rs = np.random.RandomState(seed)
    sd = {}
    for name, t in module.state_dict().items():
        shape = tuple(t.shape)
        if name.endswith('num_batches_tracked'):
            sd[name] = torch.tensor(7, dtype=torch.long)
            continue
        n = int(np.prod(shape)) if len(shape) else 1
        z = rs.standard_normal(n).astype(np.float64)
        if name.endswith('running_var'):
            w = 0.5 + np.abs(z)
        elif name.endswith('running_mean'):
            w = 0.1 * z
        elif name.endswith('bias'):
            w = 0.05 * z
        elif len(shape) == 1:
            w = 1.0 + 0.1 * z  # BatchNorm gamma
        else:
            fan_in = int(np.prod(shape[1:]))
            w = z * np.sqrt(2.0 / fan_in)
        sd[name] = torch.from_numpy(w.astype(np.float32).reshape(shape))
    return sd


def mlp_case(name):
    return next(c for c in MLP_CASES if c[0] == name)


def conv_case(name):
    return next(c for c in CONV_CASES if c[0] == name)


def build_mlp(case, network_module=None):
    if network_module is None:
        from muzero_amd import network as network_module
    name, ishape, A, P, vs, rs, H, seed = case
    net = network_module.MuZeroMLPNet(ishape, A, P, vs, rs, H)
    net.load_state_dict(seeded_state_dict(net, seed))
    net.eval()
    return net


def build_conv(case, network_module=None):
    if network_module is None:
        from muzero_amd import network as network_module
    name, kind, ishape, A, blocks, planes, vs, rs, seed = case
    if kind == 'board':
        net = network_module.MuZeroBoardGameNet(ishape, A, blocks, planes)
    else:
        net = network_module.MuZeroAtariNet(ishape, A, blocks, planes, vs, rs)
    net.load_state_dict(seeded_state_dict(net, seed))
    net.eval()
    return net


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


# ---- Philox4x32-10 exactly as muzero_amd/csrc/mz_device.h keys it (production-mode randomness of the device envs / search) ----
def philox_uniforms(seed, c1, c2, c3, n):
    """The first `n` doubles of the device stream Philox(seed, c1, c2, c3): block i has counter (i, c1, c2, c3), key
    (seed low, seed high); uniform = ((r0 >> 5) * 2^26 + (r1 >> 6)) / 2^53."""
    M = 0xFFFFFFFF
    out = []
    for i in range(n):
        c = [i & M, c1 & M, c2 & M, c3 & M]
        k0, k1 = seed & M, (seed >> 32) & M
        for _ in range(10):
            p0 = 0xD2511F53 * c[0]
            p1 = 0xCD9E8D57 * c[2]
            c = [((p1 >> 32) ^ c[1] ^ k0) & M, p1 & M, ((p0 >> 32) ^ c[3] ^ k1) & M, p0 & M]
            k0 = (k0 + 0x9E3779B9) & M
            k1 = (k1 + 0xBB67AE85) & M
        out.append(((c[0] >> 5) * 67108864.0 + (c[1] >> 6)) / 9007199254740992.0)
    return np.array(out)
