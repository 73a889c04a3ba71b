"""A float64 reference that takes the SAME branches as the HIP learner's forward pass (round 6).

The conv / Atari learner's gradients cannot be held to a tight bar against plain float64 autograd on random weights: some pre-activation of
a full-size batch always lies within float32 rounding of zero, the float32 pass takes the other ReLU branch there, and every upstream tensor
moves by that element's contribution (tests/test_gpu_atari_learner.py: the 8e-2 / 0.25 bars).  That noise is a property of comparing two
passes that DECIDE differently, not of the kernels.  Here the reference is told what the HIP pass decided:

  * every ReLU mask of the HIP forward pass, read back through the library's diagnostic hook `mzl_debug_tensor` (csrc/learner_conv.hip):
    materialised post-ReLU tensors give `x > 0` directly; where the kernels never materialise the activation (the first ReLU of a residual
    block: relu(a y + b) is formed while the next conv stages its input, mz_learn_conv.h `IN_BNRELU`) the mask is the sign of a y + b from the
    saved raw conv output y and the BatchNorm coefficients (a, b) the kernels applied -- they evaluate fmaf(a, y, b), one rounding, so its
    sign is the sign of the exact a y + b, which float64 gives;
  * every arg-min / arg-max channel of `normalize_hidden_state` (util.py:31-36), first attaining channel as in k_lc_entry.

`torch.nn.functional.relu` and `muzero_amd.network.normalize_hidden_state` are replaced, for the duration of one float64 pass of
`learner.loss_tensors`, by versions that consume those decisions in call order (shapes are checked at every site, so a wrong order cannot
go unnoticed).  What is compared afterwards is float32 arithmetic against float64 arithmetic ON THE SAME PIECEWISE-LINEAR BRANCH: an indexing,
tap, halo, tile-border or mask error of the kernels is still a full-size error; rounding is 1e-6.  Test infrastructure only."""
import copy
import ctypes as C

import numpy as np
import torch


class HipTensors:
    """Read-back of the conv learner's saved tensors (device -> host, after a synchronise)."""

    def __init__(self, hl):
        from muzero_amd.hip_learner import load_library

        self.hl, self.lib = hl, load_library()
        self.lib.mzl_debug_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        self.lib.mzl_debug_tensor.restype = C.c_int
        self.hip = C.CDLL('libamdhip64.so')
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        torch.cuda.synchronize()

    def get(self, what, a, b, shape):
        ptr, cnt = C.c_void_p(), C.c_int64()
        rc = self.lib.mzl_debug_tensor(self.hl._h, what.encode(), int(a), int(b), C.byref(ptr), C.byref(cnt))
        assert rc == 0, (what, a, b)
        n = int(np.prod(shape))
        assert n <= cnt.value, (what, a, b, shape, cnt.value)
        out = np.empty(n, np.float32)
        assert self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), ptr, n * 4, 2) == 0  # hipMemcpyDeviceToHost
        return out.reshape(shape)


def _fma_sign(y, coef, Cn, cpad):
    """mask of relu(fmaf(a, y, b)): the sign of the exact a y + b (see the module docstring); y [B, C, h, w] float32, coef the kernels' [3][cpad]"""
    a = coef[:Cn].astype(np.float64).reshape(1, Cn, 1, 1)
    b = coef[cpad:cpad + Cn].astype(np.float64).reshape(1, Cn, 1, 1)
    return (a * y.astype(np.float64) + b) > 0.0


def hip_decisions(hl, net, B, K):
    """(relu masks, normalisation index pairs) of the HIP learner's LAST forward pass, in the call order of `learner.loss_tensors` on
    `muzero_amd.network`'s modules (network.py: represent -> [prediction, dynamics] x K)."""
    T = HipTensors(hl)
    spec = net.planner_spec()
    P, R, A = spec['num_planes'], spec['num_res_blocks'], spec['num_actions']
    atari = spec['kind'] == 'atari'
    h, w = (6, 6) if atari else spec['input_shape'][1:]
    cpad = (P + 15) // 16 * 16
    masks, norms = [], []

    def tower(app, conv0, Rn, hh, ww):
        li = 0
        if conv0:  # conv + BatchNorm + ReLU in front of the blocks: materialised by k_lc_apply
            masks.append(T.get('x', app, 0, (B, P, hh, ww)) > 0)
            li = 1
        for r in range(Rn):
            y1 = T.get('y', app, li + 2 * r, (B, P, hh, ww))
            masks.append(_fma_sign(y1, T.get('fcoef', app, li + 2 * r, (3 * cpad,)), P, cpad))  # relu(bn(conv1)): never materialised
            masks.append(T.get('x', app, li + r, (B, P, hh, ww)) > 0)                           # the block's output relu(bn(conv2) + x)
        return T.get('x', app, li + Rn - 1, (B, P, hh, ww)) if Rn else None

    def norm_of(x):  # first attaining channel (k_lc_entry: ascending channels, strict comparisons); numpy's argmin / argmax are first-occurrence too
        norms.append((x.argmin(axis=1), x.argmax(axis=1)))

    if atari:
        H1, H2, H3 = spec['input_shape'][1] // 2, spec['input_shape'][1] // 4, spec['input_shape'][1] // 8
        masks.append(T.get('a1', 0, 0, (B, 128, H1, H1)) > 0)
        for r in range(2):  # (the block's inner ReLU exists as tiles only: its mask is the sign of a y + b, as in the towers)
            masks.append(_fma_sign(T.get('s48_y1', 0, r, (B, 128, H1, H1)), T.get('s48_fcoef1', 0, r, (3 * 128,)), 128, 128))
            masks.append(T.get('s48_x', 0, r, (B, 128, H1, H1)) > 0)
        masks.append(T.get('a2', 0, 0, (B, P, H2, H2)) > 0)
        for r in range(2):
            masks.append(_fma_sign(T.get('s24_y1', 0, r, (B, P, H2, H2)), T.get('s24_fcoef1', 0, r, (3 * cpad,)), P, cpad))
            masks.append(T.get('s24_x', 0, r, (B, P, H2, H2)) > 0)
        tower(2 * K + 1, False, 2, H3, H3)
        norm_of(T.get('hraw', 0, 0, (B, P, h, w)))
    else:
        norm_of(tower(0, True, R, h, w))
    feat = T.get('feat', 0, 0, (3 * K, B, 2, h * w))
    for t in range(K):
        tower(1 + K + t, False, R, h, w)                                  # prediction tower
        masks.append(feat[3 * t + 1].reshape(B, 2, h, w) > 0)              # policy head (two planes)
        masks.append(feat[3 * t + 2][:, :1].reshape(B, 1, h, w) > 0)       # value head
        xd = tower(1 + t, True, R, h, w)                                   # dynamics tower
        masks.append(feat[3 * t][:, :1].reshape(B, 1, h, w) > 0)           # reward head
        norm_of(xd)
    return masks, norms


class ForcedDecisions:
    """Context manager: while active, F.relu and network.normalize_hidden_state consume the given decisions in call order."""

    def __init__(self, masks, norms, device):
        self.masks, self.norms, self.dev = list(masks), list(norms), device
        self.i = self.j = 0
        self.flipped = 0  # ReLU sites where the float64 pre-activation's own sign disagrees with the forced mask (the kinks the plain comparison trips on)

    def __enter__(self):
        import torch.nn.functional as F

        from muzero_amd import network as nw

        self._F, self._relu, self._nw, self._norm = F, F.relu, nw, nw.normalize_hidden_state
        me = self

        def relu(x, inplace=False):
            m = me.masks[me.i]
            me.i += 1
            assert tuple(x.shape) == tuple(m.shape), ('ReLU site %d: the float64 pass has %s here, the HIP pass %s' % (me.i - 1, tuple(x.shape), m.shape))
            mt = torch.from_numpy(m).to(me.dev)
            me.flipped += int(((x.detach() > 0) != mt).sum())
            return x * mt.to(x.dtype)

        def normalize(hs):
            imn, imx = me.norms[me.j]
            me.j += 1
            assert tuple(hs.shape[0:1] + hs.shape[2:]) == tuple(imn.shape), (tuple(hs.shape), imn.shape)
            imn_t = torch.from_numpy(imn).to(me.dev).unsqueeze(1)
            imx_t = torch.from_numpy(imx).to(me.dev).unsqueeze(1)
            mn, mx = hs.gather(1, imn_t), hs.gather(1, imx_t)
            return (hs - mn) / (mx - mn + 1e-8)

        F.relu = relu
        nw.normalize_hidden_state = normalize
        return self

    def __exit__(self, *exc):
        self._F.relu = self._relu
        self._nw.normalize_hidden_state = self._norm
        if exc[0] is None:
            assert self.i == len(self.masks) and self.j == len(self.norms), (self.i, len(self.masks), self.j, len(self.norms))


def forced_f64(net, tr, w, dev, masks, norms, dtype=torch.float64):
    """float64 autograd of `learner.loss_tensors` on a copy of `net` with the given decisions forced; returns (loss, priorities, gradients by
    parameter name, number of flipped ReLU decisions).  dtype=torch.float32: PyTorch-ROCm's own float32 arithmetic on the same branch -- the
    yardstick for what float32 can reach on a batch (deep residual towers in train-mode BatchNorm amplify rounding)."""
    from muzero_amd import learner

    net_d = copy.deepcopy(net).to(dtype)
    net_d.train()
    t = lambda x, dt: torch.from_numpy(np.asarray(x)).to(dev).to(dt)  # noqa: E731
    with ForcedDecisions(masks, norms, dev) as fd:
        loss, prio = learner.loss_tensors(net_d, t(tr.state, dtype), t(tr.action, torch.int64), t(tr.value, dtype), t(tr.reward, dtype), t(tr.pi_prob, dtype), t(w, dtype))
    loss.backward()
    return float(loss.detach()), prio.detach(), {k: p.grad for k, p in net_d.named_parameters()}, fd.flipped


def tensor_errors(gd, views):
    """Per gradient tensor: max |difference| over the tensor's largest entry -- but no finer than 1e-2 of the largest entry of its network part
    (representation / dynamics / prediction): a tensor whose exact gradient nearly cancels (a tower's last BatchNorm shift, cancelled by the
    next layer's batch statistics; the one-element bias of an MSE head on a balanced batch: 2e-5 beside entries of 0.2) is measured against the
    gradients around it -- 1e-4 of that floor is 1e-6 of the part's largest gradient, a dozen float32 ulps."""
    part_max = {}
    for k, g in gd.items():
        part = k.split('.')[0]
        part_max[part] = max(part_max.get(part, 0.0), float(g.abs().max()))
    errs = {}
    for k, g in gd.items():
        pm = part_max[k.split('.')[0]]
        scale = pm if g.numel() == 1 else max(float(g.abs().max()), 1e-2 * pm)  # (a one-element tensor -- an MSE head's bias -- has no 'largest entry' of its own)
        e = float((g - views[k].double()).abs().max()) / max(scale, 1e-300)
        # BatchNorm shifts -- sums of dz over every position of every image, the most cancellation-prone tensors of the network -- count at a fifth
        # (the rule of tests/test_gpu_atari_learner.grad_errors since round 5; measured 3.4e-4 on the 48 x 48 stage of a 3-image batch)
        errs[k] = e / 5 if k.endswith('.1.bias') else e
    return errs
