"""Import shim used ONLY by oracle/gen_golden.py in the build container.

The reference (michaelnny/muzero, mounted read-only at /root/reference) is pure Python but
imports third-party packages that are absent from this image (absl, gym, six, snappy, cv2,
tensorboard).  None of them does arithmetic on the planning path; they are registered here as
empty placeholder modules so that `import muzero.mcts / network / util / pipeline / games.*`
succeeds and the *reference's own code* can be executed to record golden vectors.

Nothing in this file travels to the GPU box as a dependency of tests, smoke() or bench.py:
the fixtures written by gen_golden.py are plain .npz data.
"""
import sys
import types

import numpy as np

REFERENCE_ROOT = '/root/reference'


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    if getattr(install, '_done', False):
        return
    install._done = True

    # numpy 2 removed np.bool8 (used at gym_env.py:365, games/env.py:80,103).
    if not hasattr(np, 'bool8'):
        np.bool8 = np.bool_

    # absl: only logging/flags/app names are touched at import time.
    class _Logging:
        INFO = 20
        _warn_preinit_stderr = 0

        def __getattr__(self, _):
            return lambda *a, **k: None

    _module('absl', logging=_Logging(), flags=types.SimpleNamespace(FLAGS=None), app=types.SimpleNamespace())
    sys.modules['absl.logging'] = sys.modules['absl'].logging
    sys.modules['absl.flags'] = sys.modules['absl'].flags
    sys.modules['absl.app'] = sys.modules['absl'].app

    # gym: Env base class and the two space containers (shape/n holders only).
    class Env:
        metadata = {}

        def reset(self, **kwargs):
            return None

        def close(self):
            return None

    class Wrapper(Env):
        # gym.Wrapper (gym 0.23.1 core.py): attribute access and reset / step / close are forwarded to the wrapped env
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            return getattr(self.env, name)

        def reset(self, **kwargs):
            return self.env.reset(**kwargs)

        def step(self, action):
            return self.env.step(action)

        def close(self):
            return self.env.close()

    class ObservationWrapper(Wrapper):
        # gym.ObservationWrapper: reset / step outputs pass through self.observation()
        def reset(self, **kwargs):
            return self.observation(self.env.reset(**kwargs))

        def step(self, action):
            observation, reward, done, info = self.env.step(action)
            return self.observation(observation), reward, done, info

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Discrete:
        def __init__(self, n):
            self.n = n

    spaces = _module('gym.spaces', Box=Box, Discrete=Discrete)
    _module(
        'gym',
        Env=Env,
        Wrapper=Wrapper,
        ObservationWrapper=ObservationWrapper,
        RewardWrapper=Wrapper,
        spaces=spaces,
        wrappers=types.SimpleNamespace(),
    )

    import io

    _module('six', StringIO=io.StringIO)
    _module('snappy', compress=lambda b: bytes(b), uncompress=lambda b: bytes(b))
    _module('cv2')

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, _):
            return lambda *a, **k: None

    import torch.utils  # noqa: F401

    _module('torch.utils.tensorboard', SummaryWriter=SummaryWriter)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
