"""Round 6 diagnostic (stamps build): how far apart do the workgroups of one search launch finish?  A launch ends with its slowest workgroup;
the envs of different workgroups are independent.  python -m muzero_amd.build --stamps && python tools/dev/wg_spread.py [cartpole|tictactoe|lunar]"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np

from muzero_amd import planner as pl

pl.LIB_PATH = os.environ.get('MZ_STAMPS_LIB', os.path.join(REPO, 'muzero_amd', 'lib', 'libmzplanner_hip_stamps.so'))
from helpers import build_mlp, mlp_case

g = sys.argv[1] if len(sys.argv) > 1 else 'cartpole'
board = g == 'tictactoe'
net = build_mlp(mlp_case(g))
B, S = 4096, 25 if board else 50
kw = dict(num_simulations=S, discount=1.0 if board else 0.997, is_board_game=board, known_bounds=(-1.0, 1.0) if board else None)
if g == 'lunar':
    kw.update(root_dirichlet_alpha=0.25, root_exploration_eps=0.25)
p = pl.Planner(pl.make_mz_config(net.planner_spec(), None, num_envs=B, **kw), 0)
p.load_state_dict(net.state_dict())
p.selfplay_reset(pl.ENV_TICTACTOE if board else (pl.ENV_SYNTHETIC if g == 'lunar' else pl.ENV_CARTPOLE))
T = -1.0 if board else 1.0
p.selfplay_step(T, 20)
p.lib.mz_debug_read_wg_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
rows = []
for _ in range(10):
    p.selfplay_step(T, 1)
    buf = (C.c_longlong * 2048)()
    p.lib.mz_debug_read_wg_cycles(p.h, buf)
    a = np.array(buf[:], dtype=np.float64)
    dur, start = a[:256], a[1024:1280]
    end = start + dur
    rows.append((dur.min(), dur.mean(), dur.max(), start.max() - start.min(), end.max() - start.min(), (end.max() - start.min()) / dur.mean()))
r = np.array(rows)
print(f'{g}: per-workgroup duration (cycles, stamps build) min {r[:, 0].mean():.0f} mean {r[:, 1].mean():.0f} max {r[:, 2].mean():.0f}; start skew {r[:, 3].mean():.0f}; '
      f'launch span (first start to last end) {r[:, 4].mean():.0f} = {r[:, 5].mean():.4f} x the mean workgroup')
print('   max / mean duration: %.4f;  (a persistent kernel over many moves would run at the MEAN)' % (r[:, 2] / r[:, 1]).mean())
