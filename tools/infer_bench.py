#!/usr/bin/env python3
"""Stand-alone batched inference through the C-ABI (network.py:62-111 counterparts): host-inclusive rate of
initial_inference / recurrent_inference at the C2 shapes; run it under `rocprofv3 --kernel-trace --stats` for the kernel times.
    python tools/infer_bench.py [envs=4096] [calls=20]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np  # noqa: E402

from helpers import build_mlp, mlp_case  # noqa: E402
from muzero_amd import planner as pl  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    case = mlp_case('cartpole')
    net = build_mlp(case)
    p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=B, num_simulations=50, discount=0.997), 0)
    p.load_state_dict(net.state_dict())
    rs = np.random.RandomState(0)
    obs = rs.uniform(-1, 1, size=(B,) + tuple(case[1])).astype(np.float32)
    hidden, _, _ = p.initial_inference(obs)
    act = rs.randint(0, case[2], B).astype(np.int32)
    for name, fn in (('initial_inference', lambda: p.initial_inference(obs)), ('recurrent_inference', lambda: p.recurrent_inference(hidden, act))):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        dt = (time.perf_counter() - t0) / n
        print(f'{name}: {B} envs in {1e3 * dt:.3f} ms per call incl. host copies = {B / dt / 1e6:.2f} M inferences/s')


if __name__ == '__main__':
    main()
