"""MZLC_TWO_STREAMS=1 (the two towers of a step as two chains on two streams) against MZLC_NO_PAIR=1 (one job per launch, one stream): the same
kernels and sums, so loss, priorities, gradient and running statistics must be the same bits."""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from helpers import build_conv, conv_case  # noqa: E402
from test_gpu_conv_learner import _batch, _hip, _net, _ring  # noqa: E402

dev = torch.device('cuda', 0)
rs = np.random.RandomState(3)
for name in ('board15', 'board6', 'atari_m'):
    if name == 'atari_m':
        net = build_conv(conv_case('atari_m')).to(dev)
        A, B, K, shape = 4, 3, 5, (8, 96, 96)
    else:
        side = 15 if name == 'board15' else 6
        net, A = _net(side, 32, 2, 3, 7, dev)
        B, K, shape = 24, 5, (3, side, side)
    net.train()
    tr = _batch(rs, B, shape, A, K=K)
    w = torch.from_numpy(rs.uniform(0.3, 1.0, B).astype(np.float32)).to(dev)
    out = []
    for flag in ('MZLC_NO_PAIR', 'MZLC_TWO_STREAMS'):
        os.environ[flag] = '1'
        try:
            hl = _hip(copy.deepcopy(net), dev, B, K=K)
        finally:
            del os.environ[flag]
        for rep in range(3):
            la, pa = hl.grad(_ring(tr, dev), None, w, B)
        torch.cuda.synchronize()
        out.append((la.clone(), pa.clone(), hl.grad_flat.clone(), hl.running.clone()))
        hl.close()
    same = all(torch.equal(a, b) for a, b in zip(out[0], out[1]))
    print(name, 'bit-identical' if same else 'DIFFERENT', float((out[0][2] - out[1][2]).abs().max()))
