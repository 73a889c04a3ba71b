// learner_conv.hip -- host side of the conv-net learner step (MuZeroBoardGameNet): layer tables, saved-tensor plan, the launch schedule of one update
// over the kernels of mz_learn_conv.h.  Called by learner.hip for net_kind == MZL_NET_BOARD; same rules as the MLP path: one handle == one GPU,
// every call only enqueues on the caller's stream, no allocation after create, no CPU fallback.
//
// Schedule of mzl_grad (reference: pipeline.py:575-592 unroll, :594-609 loss / priorities; network.py:273-299, 356-498):
//   forward   representation tower -> normalize -> for t < K: [dynamics tower_t || prediction tower_t] (independent: launched PAIRED, two jobs per
//             kernel) -> normalize;  every conv writes its raw output y and the partial sums of its BatchNorm statistics, the NEXT conv applies
//             relu(a y + b) while staging; only block outputs (needed by the residual) are materialised
//   heads     all 3 K head applications side by side: 1x1 conv, batch statistics, logits, losses, priorities, backward to the tower outputs
//   backward  for t = K-1 .. 0: gradient entry (normalize backward x 0.5 + reward-head gradient | policy + value gradient), then the two towers
//             PAIRED: per conv layer BatchNorm-backward coefficients -> weight gradient (partials per image chunk + ordered reduction) -> data
//             gradient with the next mask / partial sums in its epilogue; last the representation tower
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mz_learn_conv.h"
#include "mz_learn_conv_host.h"

using namespace mzlc;

namespace {

struct BnInfo {
    int C = 0, cpad = 0, gamma_off = 0, beta_off = 0, rm_off = 0, rv_off = 0, nbt = 0;
    std::string prefix;
};
struct LayerInfo {
    int cin_real = 0, cin = 0, cout = 0, w_off = 0, cin_d = 0;
    int f_off = 0, d_off = 0, n_cb = 0, co_tiles = 0, n_cb_d = 0, co_tiles_d = 0;
    BnInfo bn;
};
struct TowerInfo {
    std::vector<int> layers;  // [conv0], then 2 R block convs
    bool conv0 = false;
};
struct TensorInfo {
    std::string name;
    int64_t off;
    int rows, cols;
};
struct AppBufs {  // saved tensors of ONE application of a tower (representation: 1, dynamics / prediction: K)
    std::vector<float*> y, x, fcoef, save, bcoef;
};
enum OpKind { OP_CONV, OP_WGRAD, OP_WREDUCE, OP_BNFWD, OP_BNBWD, OP_APPLY };
struct Op {
    int kind;
    LcConv conv;
    LcWgrad wg;
    LcWreduce wr;
    LcBnFwd bf;
    LcBnBwd bb;
    LcApply ap;
};

int pad16(int x) { return (x + 15) & ~15; }
int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

struct mzlc_learner {
    mzl_config cfg{};
    int device = 0, num_cus = 256;
    int P = 0, C0 = 0, A = 0, R = 0, K = 0, h = 0, w = 0, hw = 0, maxB = 0;
    int npt = 15, G = 1, qstride = 0;
    bool xcd_remap = true;   // k_lc_wgrad: the blocks of one image chunk on one XCD (MZLC_NO_XCD_REMAP=1 at create: launch order)
    bool fuse_apply = true;  // block outputs formed in the next conv's staging (MZLC_NO_FUSE_APPLY=1 at create: one k_lc_apply per block)
    bool side15 = false;  // the 15 x 15 build of the conv kernel (geometry as compile-time constants); MZLC_NO_SIDE=1 at create: the generic build
    int P4 = 0, nsteps = 0, SPY = 0, SPX = 0;
    std::vector<LayerInfo> layers;
    TowerInfo tower[3];  // 0 representation, 1 dynamics, 2 prediction
    std::vector<TensorInfo> tensors;
    std::vector<BnInfo> bns;  // every BatchNorm (towers, then heads): buffer table
    LchHead head[3];
    int64_t total = 0, nrunning = 0;
    bool paired = true;  // the two towers of an unroll step in paired launches (MZLC_NO_PAIR=1 at create: one job per launch; same results)
    int lastB = 0;
    std::vector<void*> allocs;
    float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr, *running = nullptr;
    int64_t* nbt = nullptr;
    bool committed = false;
    // device buffers
    float* packed = nullptr;
    size_t packed_floats = 0;
    LcPackJob* d_pack = nullptr;
    int n_pack = 0;
    float* lwT = nullptr;
    int lwT_off[3] = {0, 0, 0};
    size_t T = 0;  // floats of one activation tensor at max_batch
    AppBufs app_rep;
    std::vector<AppBufs> app_dyn, app_pred;
    std::vector<float*> s;  // normalised hidden states s_0 .. s_{K-1}
    float* obs = nullptr;
    int* act = nullptr;
    float* stat[2] = {nullptr, nullptr};
    int stat_groups_cap = 0;
    float* wpart[2] = {nullptr, nullptr};
    float* D[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    float *GsA = nullptr, *GsB = nullptr, *GsP = nullptr;
    float *dF_pred = nullptr, *dF_rew = nullptr;
    // heads
    LchGroup* d_groups = nullptr;
    std::vector<LchGroup> groups_host;
    float *hu = nullptr, *hdz = nullptr, *hfeat = nullptr, *hdl = nullptr, *hspart = nullptr, *hcoef = nullptr, *hsave = nullptr, *hlpart = nullptr, *hwpart = nullptr;
    int hp_off[3] = {0, 0, 0}, hp_total = 0;
    int n_max = 1;
    float* d_sq = nullptr;
    int sq_blocks = 0;
    const void* checked_ptr[9] = {};
};

namespace {

template <typename T>
hipError_t dalloc(mzlc_learner* h, T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T) + 256);
    if (e != hipSuccess) return e;
    e = hipMemset(q, 0, count * sizeof(T) + 256);
    h->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return e;
}

void add_tensor(mzlc_learner* h, const std::string& name, int rows, int cols, int* off_out) {
    *off_out = (int)h->total;
    h->tensors.push_back(TensorInfo{name, h->total, rows, cols});
    h->total += (int64_t)rows * (cols ? cols : 1);
}

BnInfo add_bn(mzlc_learner* h, const std::string& prefix, int C) {
    BnInfo b;
    b.C = C; b.cpad = pad16(C); b.prefix = prefix;
    add_tensor(h, prefix + ".weight", C, 0, &b.gamma_off);
    add_tensor(h, prefix + ".bias", C, 0, &b.beta_off);
    b.rm_off = (int)h->nrunning; b.rv_off = b.rm_off + C;
    h->nrunning += 2 * C;
    b.nbt = (int)h->bns.size();
    h->bns.push_back(b);
    return b;
}

int add_conv(mzlc_learner* h, const std::string& conv_name, const std::string& bn_prefix, int cin_real, int n_act, int cout, int cin_d) {
    LayerInfo L;
    L.cin_real = cin_real; L.cin = cin_real + n_act; L.cout = cout; L.cin_d = cin_d;
    add_tensor(h, conv_name + ".weight", cout, L.cin * 9, &L.w_off);
    L.bn = add_bn(h, bn_prefix, cout);
    L.n_cb = cdiv(L.cin, 16); L.co_tiles = cdiv(cout, 16);
    L.n_cb_d = cdiv(cout, 16); L.co_tiles_d = cin_d ? cdiv(cin_d, 16) : 0;
    L.f_off = (int)h->packed_floats;
    h->packed_floats += (size_t)L.co_tiles * L.n_cb * 9 * 256;
    L.d_off = (int)h->packed_floats;
    h->packed_floats += (size_t)L.co_tiles_d * L.n_cb_d * 9 * 256;
    h->layers.push_back(L);
    return (int)h->layers.size() - 1;
}

void add_tower(mzlc_learner* h, int ti, const std::string& net, bool conv0, int cin0_real, int n_act, bool input_needs_grad) {
    TowerInfo& t = h->tower[ti];
    t.conv0 = conv0;
    const int P = h->P;
    if (conv0) t.layers.push_back(add_conv(h, net + ".conv_block.0", net + ".conv_block.1", cin0_real, n_act, P, input_needs_grad ? cin0_real : 0));
    for (int r = 0; r < h->R; r++) {
        const std::string b = net + ".res_blocks." + std::to_string(r);
        t.layers.push_back(add_conv(h, b + ".conv_block1.0", b + ".conv_block1.1", P, 0, P, P));
        t.layers.push_back(add_conv(h, b + ".conv_block2.0", b + ".conv_block2.1", P, 0, P, P));
    }
}

void add_head(mzlc_learner* h, int hi, const std::string& name, int oc, int n_out, int kind) {
    LchHead& H = h->head[hi];
    H.oc = oc; H.n_out = n_out; H.kind = kind;
    add_tensor(h, name + ".0.weight", oc, h->P, &H.w1_off);
    const BnInfo b = add_bn(h, name + ".1", oc);
    H.gamma_off = b.gamma_off; H.beta_off = b.beta_off; H.rm_off = b.rm_off; H.rv_off = b.rv_off; H.nbt_idx = b.nbt;
    add_tensor(h, name + ".4.weight", n_out, oc * h->hw, &H.lw_off);
    add_tensor(h, name + ".4.bias", n_out, 0, &H.lb_off);
}

bool alloc_app(mzlc_learner* h, const TowerInfo& t, AppBufs& a) {
    bool ok = true;
    const int nl = (int)t.layers.size(), nx = h->R + (t.conv0 ? 1 : 0);
    a.y.resize(nl); a.fcoef.resize(nl); a.save.resize(nl); a.bcoef.resize(nl); a.x.resize(nx);
    const int cpad = pad16(h->P);
    for (int i = 0; i < nl; i++) {
        ok = ok && dalloc(h, &a.y[i], h->T) == hipSuccess && dalloc(h, &a.fcoef[i], (size_t)3 * cpad) == hipSuccess &&
             dalloc(h, &a.save[i], (size_t)2 * cpad) == hipSuccess && dalloc(h, &a.bcoef[i], (size_t)3 * cpad) == hipSuccess;
    }
    for (int i = 0; i < nx; i++) ok = ok && dalloc(h, &a.x[i], h->T) == hipSuccess;
    return ok;
}

size_t conv_lds(const mzlc_learner* h, int cpad_in) { return ((size_t)8 * h->qstride + (size_t)3 * cpad_in) * sizeof(float); }
size_t wgrad_lds(const mzlc_learner* h) { return ((size_t)32 * (h->SPY + h->SPX) + 160) * sizeof(float); }

// ---- op builders -------------------------------------------------------------------------------------------------------------
struct Sched {
    mzlc_learner* h;
    int B, lane;
    bool lane_pairs;  // this tower's launches are paired with another tower's
    int groups() const { return cdiv(B, h->G); }

    LcConv conv_base(const LayerInfo& L, bool dgrad) const {
        LcConv c{};
        c.B = B; c.G = h->G; c.h = h->h; c.w_img = h->w; c.qstride = h->qstride;
        if (!dgrad) {
            c.cin_real = L.cin_real; c.cin = L.cin; c.n_cb = L.n_cb; c.cout = L.cout; c.co_tiles = L.co_tiles;
            c.w = h->packed + L.f_off;
        } else {
            c.cin_real = L.cout; c.cin = L.cout; c.n_cb = L.n_cb_d; c.cout = L.cin_d; c.co_tiles = L.co_tiles_d;
            c.w = h->packed + L.d_off;
        }
        c.cpad_in = pad16(c.cin_real); c.cpad_out = pad16(c.cout);
        c.num_actions = h->A;
        return c;
    }
    Op op_conv(const LcConv& c) const { Op o{}; o.kind = OP_CONV; o.conv = c; return o; }
    Op op_bnfwd(const LayerInfo& L, float* fcoef, float* save) const {
        Op o{};
        o.kind = OP_BNFWD;
        LcBnFwd& f = o.bf;
        f.part = h->stat[lane]; f.gamma = h->params + L.bn.gamma_off; f.beta = h->params + L.bn.beta_off; f.coef = fcoef; f.save = save;
        f.running_mean = h->running ? h->running + L.bn.rm_off : nullptr; f.running_var = h->running ? h->running + L.bn.rv_off : nullptr;
        f.num_batches = h->nbt ? h->nbt + L.bn.nbt : nullptr;
        f.groups = groups(); f.C = L.bn.C; f.cpad = L.bn.cpad; f.count = (float)B * (float)h->hw;
        return o;
    }
    Op op_bnbwd(const LayerInfo& L, const float* save, float* bcoef, int ngroups, int accumulate) const {
        Op o{};
        o.kind = OP_BNBWD;
        LcBnBwd& f = o.bb;
        f.part = h->stat[lane]; f.gamma = h->params + L.bn.gamma_off; f.save = save; f.coef = bcoef;
        f.dgamma = h->grads + L.bn.gamma_off; f.dbeta = h->grads + L.bn.beta_off;
        f.groups = ngroups; f.C = L.bn.C; f.cpad = L.bn.cpad; f.accumulate = accumulate; f.count = (float)B * (float)h->hw;
        return o;
    }
    Op op_apply(const float* y, const float* res, const float* coef, float* out) const {
        Op o{};
        o.kind = OP_APPLY;
        o.ap.y = y; o.ap.res = res; o.ap.coef = coef; o.ap.out = out; o.ap.C = h->P; o.ap.hw = h->hw; o.ap.cpad = pad16(h->P);
        o.ap.n = (long long)B * h->P * h->hw;
        return o;
    }
    void wgrad_ops(std::vector<Op>& ops, const LayerInfo& L, const float* dz, const float* y, const float* bcoef, const float* x0, int x_mode, const float* xcoef,
                   const int* action, int accumulate) const {
        Op o{};
        o.kind = OP_WGRAD;
        LcWgrad& g = o.wg;
        g.dz = dz; g.y = y; g.dcoef = bcoef; g.x0 = x0; g.xcoef = xcoef; g.x_mode = x_mode; g.action = action; g.num_actions = h->A;
        g.cin_real = L.cin_real; g.cin = L.cin; g.cout = L.cout; g.ci_tiles = cdiv(L.cin, 16); g.co_tiles = L.co_tiles;
        g.cpad_in = pad16(L.cin_real); g.cpad_out = pad16(L.cout);
        g.B = B; g.h = h->h; g.w_img = h->w; g.P4 = h->P4; g.nsteps = h->nsteps; g.SPY = h->SPY; g.SPX = h->SPX;
        g.co_blocks = cdiv(g.co_tiles, 2);
        const int ci_blocks = cdiv(g.ci_tiles, 2);
        // two workgroups per CU in all: a paired launch brings the other half; the first conv blocks (action planes: the dynamics tower's extra
        // ops) are never paired
        int chunks = ((lane_pairs && !action) ? 1 : 2) * h->num_cus / (g.co_blocks * ci_blocks);
        chunks = chunks < 1 ? 1 : (chunks > B ? B : chunks);
        g.ipw = cdiv(B, chunks);
        chunks = cdiv(B, g.ipw);
        g.part = h->wpart[lane];
        ops.push_back(o);
        Op r{};
        r.kind = OP_WREDUCE;
        r.wr.part = g.part; r.wr.grad = h->grads + L.w_off; r.wr.chunks = chunks; r.wr.cout = L.cout; r.wr.cin = L.cin;
        r.wr.co_pad = g.co_tiles * 16; r.wr.ci_pad = g.ci_tiles * 16; r.wr.accumulate = accumulate;
        ops.push_back(r);
    }

    // forward of one tower application; returns the tower's output tensor
    float* tower_fwd(std::vector<Op>& ops, const TowerInfo& t, AppBufs& a, const float* x_in, const int* action) const {
        const float* cur = x_in;
        int li = 0, xi = 0;
        if (t.conv0) {
            const LayerInfo& L = h->layers[t.layers[0]];
            LcConv c = conv_base(L, false);
            c.in0 = cur; c.in_mode = IN_IDENT; c.action = action; c.out = a.y[0]; c.stat_mode = ST_FWD; c.stat_part = h->stat[lane];
            ops.push_back(op_conv(c));
            ops.push_back(op_bnfwd(L, a.fcoef[0], a.save[0]));
            ops.push_back(op_apply(a.y[0], nullptr, a.fcoef[0], a.x[0]));
            cur = a.x[0];
            li = 1; xi = 1;
        }
        // A block's output x' = relu(a2 y2 + b2 + x) is formed by the NEXT block's first conv while it stages its input (IN_BNRES) and written
        // through once (mat_out): the residual, the weight gradient and the ReLU masks read it later.  Only the tower's last block output has no
        // consumer conv: k_lc_apply materialises it.  (h->fuse_apply == false: one k_lc_apply per block, the round's first form; same bits.)
        const float* pend_y = nullptr;      // y2 of the previous block, if its output has not been materialised yet
        const float* pend_coef = nullptr;
        const float* pend_res = nullptr;
        float* pend_out = nullptr;
        for (int r = 0; r < h->R; r++) {
            const int l1 = li + 2 * r, l2 = l1 + 1;
            const LayerInfo &L1 = h->layers[t.layers[l1]], &L2 = h->layers[t.layers[l2]];
            LcConv c = conv_base(L1, false);
            if (pend_y) {
                c.in0 = pend_y; c.in1 = pend_res; c.coef = pend_coef; c.in_mode = IN_BNRES; c.mat_out = pend_out;
                cur = pend_out;
                pend_y = nullptr;
            } else {
                c.in0 = cur; c.in_mode = IN_IDENT;
            }
            c.out = a.y[l1]; c.stat_mode = ST_FWD; c.stat_part = h->stat[lane];
            ops.push_back(op_conv(c));
            ops.push_back(op_bnfwd(L1, a.fcoef[l1], a.save[l1]));
            LcConv d = conv_base(L2, false);
            d.in0 = a.y[l1]; d.in_mode = IN_BNRELU; d.coef = a.fcoef[l1]; d.out = a.y[l2]; d.stat_mode = ST_FWD; d.stat_part = h->stat[lane];
            ops.push_back(op_conv(d));
            ops.push_back(op_bnfwd(L2, a.fcoef[l2], a.save[l2]));
            if (h->fuse_apply && r + 1 < h->R) {
                pend_y = a.y[l2]; pend_coef = a.fcoef[l2]; pend_res = cur; pend_out = a.x[xi + r];
            } else {
                ops.push_back(op_apply(a.y[l2], cur, a.fcoef[l2], a.x[xi + r]));
                cur = a.x[xi + r];
            }
        }
        return const_cast<float*>(cur);
    }

    // backward of one tower application.  D[lane][0] holds dz of the tower's last BatchNorm, its partial sums are in stat[lane] (entry_groups groups).
    // final_out: where the gradient wrt the tower's input goes (null: not needed -- the representation tower); final_skip: added to it.
    void tower_bwd(std::vector<Op>& ops, const TowerInfo& t, AppBufs& a, const float* x_in, const int* action, int entry_groups, int accumulate,
                   float* final_out, const float* final_skip) const {
        float *Da = h->D[lane][0], *Db = h->D[lane][1], *Dc = h->D[lane][2];
        const int li = t.conv0 ? 1 : 0, xi = t.conv0 ? 1 : 0;
        int ng = entry_groups;
        for (int r = h->R - 1; r >= 0; r--) {
            const int l1 = li + 2 * r, l2 = l1 + 1;
            const LayerInfo &L1 = h->layers[t.layers[l1]], &L2 = h->layers[t.layers[l2]];
            const float* xin_blk = r > 0 ? a.x[xi + r - 1] : (t.conv0 ? a.x[0] : x_in);
            ops.push_back(op_bnbwd(L2, a.save[l2], a.bcoef[l2], ng, accumulate));
            wgrad_ops(ops, L2, Da, a.y[l2], a.bcoef[l2], a.y[l1], IN_BNRELU, a.fcoef[l1], nullptr, accumulate);
            LcConv c = conv_base(L2, true);
            c.in0 = Da; c.in1 = a.y[l2]; c.coef = a.bcoef[l2]; c.in_mode = IN_BNBWD; c.out = Db;
            c.mask = a.y[l1]; c.mcoef = a.fcoef[l1]; c.partner = a.y[l1]; c.stat_mode = ST_BWD; c.stat_part = h->stat[lane];
            ops.push_back(op_conv(c));
            ng = groups();
            ops.push_back(op_bnbwd(L1, a.save[l1], a.bcoef[l1], ng, accumulate));
            wgrad_ops(ops, L1, Db, a.y[l1], a.bcoef[l1], xin_blk, IN_IDENT, nullptr, nullptr, accumulate);
            LcConv d = conv_base(L1, true);
            d.in0 = Db; d.in1 = a.y[l1]; d.coef = a.bcoef[l1]; d.in_mode = IN_BNBWD; d.skip = Da;
            if (r > 0 || t.conv0) {  // the block's input is itself a ReLU output behind a BatchNorm: mask + that layer's partial sums
                d.mask = xin_blk; d.mcoef = nullptr; d.partner = a.y[l1 - 1]; d.stat_mode = ST_BWD; d.stat_part = h->stat[lane];
                d.out = Dc;
            } else {
                d.stat_mode = ST_NONE;
                d.out = final_out ? final_out : Dc;
                if (final_skip) {  // (two addends: the block's skip gradient rides in `skip`, so the other tower's result goes through a second pass)
                    // not reached: the prediction tower never carries a final_skip; kept for clarity
                }
            }
            ops.push_back(op_conv(d));
            ng = groups();
            float* tmp = Da; Da = Dc; Dc = tmp;
        }
        if (t.conv0) {
            const LayerInfo& L0 = h->layers[t.layers[0]];
            ops.push_back(op_bnbwd(L0, a.save[0], a.bcoef[0], ng, accumulate));
            wgrad_ops(ops, L0, Da, a.y[0], a.bcoef[0], x_in, IN_IDENT, nullptr, action, accumulate);
            if (final_out) {
                LcConv c = conv_base(L0, true);
                c.in0 = Da; c.in1 = a.y[0]; c.coef = a.bcoef[0]; c.in_mode = IN_BNBWD; c.skip = final_skip; c.out = final_out; c.stat_mode = ST_NONE;
                ops.push_back(op_conv(c));
            }
        }
    }
};

template <int NPT, int SIDE>
void launch_conv(int mode, const Pair<LcConv>& pj, dim3 grid, size_t lds, hipStream_t st) {
    if (mode == IN_IDENT) hipLaunchKernelGGL((k_lc_conv<NPT, IN_IDENT, SIDE>), grid, dim3(256), lds, st, pj);
    else if (mode == IN_BNRELU) hipLaunchKernelGGL((k_lc_conv<NPT, IN_BNRELU, SIDE>), grid, dim3(256), lds, st, pj);
    else if (mode == IN_BNRES) hipLaunchKernelGGL((k_lc_conv<NPT, IN_BNRES, SIDE>), grid, dim3(256), lds, st, pj);
    else hipLaunchKernelGGL((k_lc_conv<NPT, IN_BNBWD, SIDE>), grid, dim3(256), lds, st, pj);
}
template <int NPT, int SIDE>
hipError_t conv_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_conv<NPT, IN_IDENT, SIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_conv<NPT, IN_BNRELU, SIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_conv<NPT, IN_BNBWD, SIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_conv<NPT, IN_BNRES, SIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
}

int launch_ops(mzlc_learner* h, const Op* a, const Op* b, hipStream_t st) {
    switch (a->kind) {
        case OP_CONV: {
            if (b && b->conv.in_mode != a->conv.in_mode) {  // (never the case for the zipped towers; kept correct anyway)
                launch_ops(h, a, nullptr, st);
                return launch_ops(h, b, nullptr, st);
            }
            Pair<LcConv> pj{};
            pj.a = a->conv;
            const int ga = cdiv(a->conv.B, a->conv.G), gb = b ? cdiv(b->conv.B, b->conv.G) : 0;
            pj.na = ga;
            if (b) pj.b = b->conv;
            int z = cdiv(a->conv.co_tiles, 4), cp = a->conv.cpad_in;
            if (b) { z = cdiv(b->conv.co_tiles, 4) > z ? cdiv(b->conv.co_tiles, 4) : z; cp = b->conv.cpad_in > cp ? b->conv.cpad_in : cp; }
            const dim3 grid(1, ga + gb, z);
            const size_t lds = conv_lds(h, cp);
            const int mode = a->conv.in_mode;
            if (h->npt == 15 && h->side15) launch_conv<15, 15>(mode, pj, grid, lds, st);
            else if (h->npt == 15) launch_conv<15, 0>(mode, pj, grid, lds, st);
            else if (h->npt == 9) launch_conv<9, 0>(mode, pj, grid, lds, st);
            else launch_conv<6, 0>(mode, pj, grid, lds, st);
            break;
        }
        case OP_WGRAD: {
            Pair<LcWgrad> pj{};
            pj.a = a->wg;
            const int ya = a->wg.co_blocks * cdiv(a->wg.B, a->wg.ipw), yb = b ? b->wg.co_blocks * cdiv(b->wg.B, b->wg.ipw) : 0;
            pj.na = ya;
            if (b) pj.b = b->wg;
            int x = cdiv(a->wg.ci_tiles, 2);
            if (b && cdiv(b->wg.ci_tiles, 2) > x) x = cdiv(b->wg.ci_tiles, 2);
            // a chunk's blocks on one XCD (k_lc_wgrad): needs the same block grid in both jobs and a group count the 8 XCDs divide
            const int groups = (ya + yb) / a->wg.co_blocks;
            pj.remap = (h->xcd_remap && (!b || (b->wg.co_blocks == a->wg.co_blocks && cdiv(b->wg.ci_tiles, 2) == cdiv(a->wg.ci_tiles, 2))) && groups % 8 == 0) ? 1 : 0;
            if (a->wg.action || (b && b->wg.action)) hipLaunchKernelGGL(k_lc_wgrad<true>, dim3(x, ya + yb), dim3(256), wgrad_lds(h), st, pj);
            else hipLaunchKernelGGL(k_lc_wgrad<false>, dim3(x, ya + yb), dim3(256), wgrad_lds(h), st, pj);
            break;
        }
        case OP_WREDUCE: {
            Pair<LcWreduce> pj{};
            pj.a = a->wr;
            const int ya = cdiv(9 * a->wr.cout * a->wr.cin, 256), yb = b ? cdiv(9 * b->wr.cout * b->wr.cin, 256) : 0;
            pj.na = ya;
            if (b) pj.b = b->wr;
            hipLaunchKernelGGL(k_lc_wreduce, dim3(1, ya + yb), dim3(256), 0, st, pj);
            break;
        }
        case OP_BNFWD: {
            Pair<LcBnFwd> pj{};
            pj.a = a->bf;
            const int ya = cdiv(a->bf.cpad, 16), yb = b ? cdiv(b->bf.cpad, 16) : 0;
            pj.na = ya;
            if (b) pj.b = b->bf;
            hipLaunchKernelGGL(k_lc_bn_fwd, dim3(1, ya + yb), dim3(256), 0, st, pj);
            break;
        }
        case OP_BNBWD: {
            Pair<LcBnBwd> pj{};
            pj.a = a->bb;
            const int ya = cdiv(a->bb.cpad, 16), yb = b ? cdiv(b->bb.cpad, 16) : 0;
            pj.na = ya;
            if (b) pj.b = b->bb;
            hipLaunchKernelGGL(k_lc_bn_bwd, dim3(1, ya + yb), dim3(256), 0, st, pj);
            break;
        }
        case OP_APPLY: {
            Pair<LcApply> pj{};
            pj.a = a->ap;
            const int ya = (int)((a->ap.n + 1023) / 1024), yb = b ? (int)((b->ap.n + 1023) / 1024) : 0;
            pj.na = ya;
            if (b) pj.b = b->ap;
            hipLaunchKernelGGL(k_lc_apply, dim3(1, ya + yb), dim3(256), 0, st, pj);
            break;
        }
        default: return -1;
    }
    return 0;
}

// two op lists of independent towers side by side; the longer one's extra ops run alone (front: forward, the dynamics tower's first conv block;
// back: backward, the same block's gradient)
int run_zip(mzlc_learner* h, const std::vector<Op>& A, const std::vector<Op>& Bv, bool extra_in_front, hipStream_t st, bool paired) {
    const size_t na = A.size(), nb = Bv.size();
    if (!paired) {  // (B first: in the backward pass the dynamics tower's last data gradient adds the prediction tower's result)
        for (const Op& o : Bv) launch_ops(h, &o, nullptr, st);
        for (const Op& o : A) launch_ops(h, &o, nullptr, st);
        return 0;
    }
    const size_t extra = na - nb;
    size_t ia = 0;
    if (extra_in_front)
        for (; ia < extra; ia++) launch_ops(h, &A[ia], nullptr, st);
    for (size_t i = 0; i < nb; i++, ia++) {
        if (A[ia].kind != Bv[i].kind) return -1;
        launch_ops(h, &A[ia], &Bv[i], st);
    }
    for (; ia < na; ia++) launch_ops(h, &A[ia], nullptr, st);
    return 0;
}

constexpr int ENTRY_SPLIT = 8;  // workgroups per image of the entry kernel (each walks every ENTRY_SPLIT-th 32-pixel chunk: 15 x 15 -> one chunk each;
                                // measured: 2 per image -- 256 workgroups looping -- is latency-bound, 90 us against 62)
int entry_groups(const mzlc_learner* h, int B) {
    const int nchunks = cdiv(h->hw, 32);
    return B * (nchunks < ENTRY_SPLIT ? nchunks : ENTRY_SPLIT);
}
void launch_entry(mzlc_learner* h, const LcEntry& e, hipStream_t st) {
    const int nchunks = cdiv(h->hw, 32);
    const dim3 grid(nchunks < ENTRY_SPLIT ? nchunks : ENTRY_SPLIT, e.B);
    const int cpt = cdiv(e.C, 8);
    if (cpt <= 2) hipLaunchKernelGGL(k_lc_entry<2>, grid, dim3(256), 0, st, e);
    else if (cpt <= 8) hipLaunchKernelGGL(k_lc_entry<8>, grid, dim3(256), 0, st, e);
    else if (cpt <= 16) hipLaunchKernelGGL(k_lc_entry<16>, grid, dim3(256), 0, st, e);
    else hipLaunchKernelGGL(k_lc_entry<32>, grid, dim3(256), 0, st, e);
}
void launch_normalize(mzlc_learner* h, const float* in, float* out, int B, hipStream_t st) {
    const dim3 grid(cdiv(h->hw, 32), B);
    const int cpt = cdiv(h->P, 8);
    if (cpt <= 2) hipLaunchKernelGGL(k_lc_normalize<2>, grid, dim3(256), 0, st, in, out, B, h->P, h->hw);
    else if (cpt <= 8) hipLaunchKernelGGL(k_lc_normalize<8>, grid, dim3(256), 0, st, in, out, B, h->P, h->hw);
    else if (cpt <= 16) hipLaunchKernelGGL(k_lc_normalize<16>, grid, dim3(256), 0, st, in, out, B, h->P, h->hw);
    else hipLaunchKernelGGL(k_lc_normalize<32>, grid, dim3(256), 0, st, in, out, B, h->P, h->hw);
}

}  // namespace

// =================================================================================================================================
int mzlc_create(const mzl_config* cfg, int device_id, int num_cus, mzlc_learner** out, std::string& err) {
    mzlc_learner* h = new mzlc_learner();
    h->cfg = *cfg; h->device = device_id; h->num_cus = num_cus > 0 ? num_cus : 256;
    h->P = cfg->num_planes; h->C0 = cfg->in_channels; h->A = cfg->num_actions; h->R = cfg->num_res_blocks; h->K = cfg->unroll_steps;
    h->h = cfg->board_h; h->w = cfg->board_w; h->hw = h->h * h->w; h->maxB = cfg->max_batch;
    h->paired = !getenv("MZLC_NO_PAIR");
    h->fuse_apply = !getenv("MZLC_NO_FUSE_APPLY");
    h->xcd_remap = !getenv("MZLC_NO_XCD_REMAP");
    auto bad = [&](const std::string& m) { err = m; mzlc_destroy(h); return MZL_E_INVALID; };
    if (h->C0 < 1 || h->h < 1 || h->w < 1 || h->R < 1 || h->P < 1) return bad("bad conv-net geometry (in_channels, board_h, board_w, num_res_blocks, num_planes)");
    if (cfg->in_dim != h->C0 * h->hw) return bad("in_dim must equal in_channels * board_h * board_w");
    if (cfg->value_support_size != 1 || cfg->reward_support_size != 1)
        return bad("the conv learner covers MuZeroBoardGameNet (squared-error value / reward heads, network.py:540-574); categorical conv heads are not built");
    if (h->hw > 240 || h->P > 1024) return bad("conv learner: boards up to 240 points and 1024 planes (larger nets train through muzero_amd.learner.train_step)");
    {   // pixel tiling of the conv kernels: G whole images per workgroup in NPT tiles of 16 pixel slots; lane = pixel quad while staging
        const int QP = (h->hw + 3) / 4;
        double best = -1.0;
        const int cand[3] = {6, 9, 15};
        for (int i = 0; i < 3; i++) {
            int G = (16 * cand[i]) / h->hw;
            if (64 / QP < G) G = 64 / QP;
            if (G < 1) continue;
            const double eff = (double)G * h->hw / (16.0 * cand[i]);
            if (eff > best + 1e-9) { best = eff; h->npt = cand[i]; h->G = G; }  // (ties: the smaller tiling -- fewer accumulators per wave)
        }
        if (best < 0.0) return bad("board does not fit the conv kernels' tiling");
        h->qstride = (4 * h->G * (h->h + 2) * (h->w + 2) + 63) & ~63;
        h->side15 = h->h == 15 && h->w == 15 && h->G == 1 && h->npt == 15 && !getenv("MZLC_NO_SIDE");
        h->P4 = 4 * cdiv(h->w + 1, 4);
        h->nsteps = cdiv(h->h * h->P4, 16);
        h->SPY = 16 * h->nsteps + 4;
        h->SPX = 2 * h->P4 + 16 * h->nsteps + 12;
        if (wgrad_lds(h) > 160 * 1024 || conv_lds(h, pad16(h->P + h->A)) > 160 * 1024) return bad("board too large for the conv learner's LDS layout");
    }
    // ---- parameter / buffer tables in state_dict order (network.py:356-498) ----
    add_tower(h, 0, "represent_net", true, h->C0, 0, false);
    add_tower(h, 1, "dynamics_net", true, h->P, h->A, true);
    add_head(h, 0, "dynamics_net.reward_head", 1, cfg->reward_support_size, 0);
    add_tower(h, 2, "prediction_net", false, 0, 0, true);
    add_head(h, 1, "prediction_net.policy_net", 2, h->A, 1);
    add_head(h, 2, "prediction_net.value_net", 1, cfg->value_support_size, 0);
    h->n_max = h->A > 1 ? h->A : 1;
    // ---- device memory ----
    h->T = (size_t)h->maxB * h->P * h->hw;
    bool ok = true;
    auto AL = [&](float** p, size_t n) { ok = ok && dalloc(h, p, n) == hipSuccess; };
    AL(&h->packed, h->packed_floats);
    {
        std::vector<LcPackJob> jobs;
        for (const LayerInfo& L : h->layers) {
            LcPackJob j{};
            j.w_off = L.w_off; j.cout = L.cout; j.cin = L.cin; j.cin_d = L.cin_d; j.f_off = L.f_off; j.d_off = L.d_off;
            j.n_cb = L.n_cb; j.co_tiles = L.co_tiles; j.n_cb_d = L.n_cb_d; j.co_tiles_d = L.co_tiles_d;
            jobs.push_back(j);
        }
        h->n_pack = (int)jobs.size();
        ok = ok && dalloc(h, &h->d_pack, jobs.size()) == hipSuccess;
        if (ok) ok = hipMemcpy(h->d_pack, jobs.data(), jobs.size() * sizeof(LcPackJob), hipMemcpyHostToDevice) == hipSuccess;
    }
    {
        size_t n = 0;
        for (int i = 0; i < 3; i++) { h->lwT_off[i] = (int)n; n += (size_t)h->head[i].n_out * h->head[i].oc * h->hw; }
        AL(&h->lwT, n);
    }
    ok = ok && alloc_app(h, h->tower[0], h->app_rep);
    h->app_dyn.resize(h->K); h->app_pred.resize(h->K);
    for (int t = 0; t < h->K; t++) ok = ok && alloc_app(h, h->tower[1], h->app_dyn[t]) && alloc_app(h, h->tower[2], h->app_pred[t]);
    h->s.resize(h->K);
    for (int t = 0; t < h->K; t++) AL(&h->s[t], h->T);
    AL(&h->obs, (size_t)h->maxB * h->C0 * h->hw);
    ok = ok && dalloc(h, &h->act, (size_t)h->K * h->maxB) == hipSuccess;
    {
        const int g_conv = cdiv(h->maxB, h->G), g_entry = h->maxB * cdiv(h->hw, 32);
        h->stat_groups_cap = g_conv > g_entry ? g_conv : g_entry;
        const int cpad = pad16(h->P);
        for (int l = 0; l < 2; l++) AL(&h->stat[l], (size_t)h->stat_groups_cap * cpad * 2);
        // weight-gradient partials: chunks <= min(B, CUs / blocks); chunks * blocks <= max(CUs, blocks)
        size_t mx = 0;
        for (const LayerInfo& L : h->layers) {
            const int cot = L.co_tiles, cit = cdiv(L.cin, 16), blocks = cdiv(cot, 2) * cdiv(cit, 2);
            int chunks = 2 * h->num_cus / blocks;
            chunks = chunks < 1 ? 1 : (chunks > h->maxB ? h->maxB : chunks);
            const size_t n = (size_t)chunks * 9 * cot * 16 * cit * 16;
            mx = n > mx ? n : mx;
        }
        for (int l = 0; l < 2; l++) AL(&h->wpart[l], mx);
        for (int l = 0; l < 2; l++)
            for (int i = 0; i < 3; i++) AL(&h->D[l][i], h->T);
    }
    AL(&h->GsA, h->T); AL(&h->GsB, h->T); AL(&h->GsP, h->T);
    AL(&h->dF_pred, h->T * h->K); AL(&h->dF_rew, h->T * h->K);
    {
        const int ng = 3 * h->K;
        ok = ok && dalloc(h, &h->d_groups, (size_t)ng) == hipSuccess;
        const size_t gb = (size_t)ng * h->maxB;
        AL(&h->hu, gb * LCH_MAXOC * h->hw); AL(&h->hdz, gb * LCH_MAXOC * h->hw); AL(&h->hfeat, gb * LCH_MAXOC * h->hw);
        AL(&h->hdl, gb * h->n_max); AL(&h->hspart, gb * LCH_MAXOC * 2); AL(&h->hcoef, (size_t)ng * LCH_MAXOC * 5); AL(&h->hsave, (size_t)ng * LCH_MAXOC * 2);
        AL(&h->hlpart, gb);
        for (int i = 0; i < 3; i++) {
            h->hp_off[i] = h->hp_total;
            h->hp_total += h->head[i].oc * h->P + h->head[i].n_out * h->head[i].oc * h->hw + h->head[i].n_out;
        }
        AL(&h->hwpart, (size_t)h->K * h->hp_total);
    }
    h->sq_blocks = (int)((h->total + 1023) / 1024);
    AL(&h->d_sq, (size_t)h->sq_blocks);
    if (!ok) {
        err = "hipMalloc failed (conv learner: " + std::to_string((double)h->T * 4 * (h->tower[0].layers.size() * 1.5 + h->K * 50) / 1e9) + " GB class)";
        mzlc_destroy(h);
        return MZL_E_HIP;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_wgrad<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_wgrad<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = conv_attr<15, 15>();
    if (e == hipSuccess) e = conv_attr<15, 0>();
    if (e == hipSuccess) e = conv_attr<9, 0>();
    if (e == hipSuccess) e = conv_attr<6, 0>();
    if (e == hipSuccess) e = hipDeviceSynchronize();  // (dalloc's fills run on the NULL stream)
    if (e != hipSuccess) {
        err = std::string("conv learner init: ") + hipGetErrorString(e);
        mzlc_destroy(h);
        return MZL_E_HIP;
    }
    *out = h;
    return MZL_OK;
}

void mzlc_destroy(mzlc_learner* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
}

int64_t mzlc_num_params(const mzlc_learner* h) { return h->total; }
int mzlc_num_tensors(const mzlc_learner* h) { return (int)h->tensors.size(); }
int mzlc_tensor_info(const mzlc_learner* h, int i, const char** name, int64_t* offset, int32_t* rows, int32_t* cols) {
    if (i < 0 || i >= (int)h->tensors.size()) return MZL_E_INVALID;
    const TensorInfo& t = h->tensors[i];
    if (name) *name = t.name.c_str();
    if (offset) *offset = t.off;
    if (rows) *rows = t.rows;
    if (cols) *cols = t.cols;
    return MZL_OK;
}
int mzlc_num_buffers(const mzlc_learner* h) { return (int)h->bns.size(); }
int mzlc_buffer_info(const mzlc_learner* h, int i, const char** name, int64_t* offset, int32_t* count) {
    if (i < 0 || i >= (int)h->bns.size()) return MZL_E_INVALID;
    const BnInfo& b = h->bns[i];
    if (name) *name = b.prefix.c_str();
    if (offset) *offset = b.rm_off;
    if (count) *count = b.C;
    return MZL_OK;
}
int64_t mzlc_num_running(const mzlc_learner* h) { return h->nrunning; }

int mzlc_bind(mzlc_learner* h, float* params, float* grads, float* m, float* v) {
    h->params = params; h->grads = grads; h->m = m; h->v = v;
    h->committed = false;
    return MZL_OK;
}
int mzlc_bind_buffers(mzlc_learner* h, float* running, int64_t* num_batches) {
    h->running = running; h->nbt = num_batches;
    return MZL_OK;
}

static int pack_all(mzlc_learner* h, hipStream_t st) {
    hipLaunchKernelGGL(k_lc_pack, dim3(64, h->n_pack), dim3(256), 0, st, h->d_pack, h->params, h->packed);
    for (int i = 0; i < 3; i++) {
        const int nf = h->head[i].oc * h->hw, n = h->head[i].n_out * nf;
        hipLaunchKernelGGL(k_lc_pack_lin, dim3(cdiv(n, 256)), dim3(256), 0, st, h->params, h->lwT + h->lwT_off[i], h->head[i].lw_off, h->head[i].n_out, nf);
    }
    return hipGetLastError() == hipSuccess ? MZL_OK : MZL_E_HIP;
}

int mzlc_commit(mzlc_learner* h, void* stream, std::string& err) {
    if (!h->params) { err = "mzl_bind first"; return MZL_E_STATE; }
    if (hipSetDevice(h->device) != hipSuccess) { err = "hipSetDevice"; return MZL_E_HIP; }
    if (pack_all(h, reinterpret_cast<hipStream_t>(stream)) != MZL_OK) { err = "pack kernels failed to launch"; return MZL_E_HIP; }
    h->committed = true;
    return MZL_OK;
}

int mzlc_grad(mzlc_learner* h, const mzl_batch* b, void* stream, std::string& err) {
    if (!h->committed) { err = "weights not committed: mzl_bind, then mzl_commit"; return MZL_E_STATE; }
    if (!h->running || !h->nbt) { err = "conv learner: mzl_bind_buffers first (BatchNorm running statistics)"; return MZL_E_STATE; }
    if (b->batch < 1 || b->batch > h->maxB) { err = "batch must be in [1, max_batch]"; return MZL_E_INVALID; }
    if (!b->d_index || !b->d_state || !b->d_action || !b->d_pi_prob || !b->d_value || !b->d_reward || !b->d_weights || !b->d_loss || !b->d_priorities) {
        err = "null batch pointer";
        return MZL_E_INVALID;
    }
    if (b->action_bytes != 1 && b->action_bytes != 2) { err = "action_bytes must be 1 or 2"; return MZL_E_INVALID; }
    if (b->action_bytes == 1 && h->A > 128) { err = "num_actions > 128 needs int16 actions"; return MZL_E_INVALID; }
    {
        const void* ptrs[9] = {b->d_state, b->d_action, b->d_pi_prob, b->d_value, b->d_reward, b->d_index, b->d_weights, b->d_loss, b->d_priorities};
        static const char* names[9] = {"d_state", "d_action", "d_pi_prob", "d_value", "d_reward", "d_index", "d_weights", "d_loss", "d_priorities"};
        for (int i = 0; i < 9; i++) {
            if (ptrs[i] == h->checked_ptr[i]) continue;
            hipPointerAttribute_t at{};
            const hipError_t e = hipPointerGetAttributes(&at, ptrs[i]);
            if (e != hipSuccess || at.type != hipMemoryTypeDevice || at.device != h->device) {
                (void)hipGetLastError();
                err = std::string(names[i]) + " is not memory of the learner's GPU (the batch is read where the replay lives: keep it in HBM)";
                return MZL_E_INVALID;
            }
            h->checked_ptr[i] = ptrs[i];
        }
    }
    if (hipSetDevice(h->device) != hipSuccess) { err = "hipSetDevice"; return MZL_E_HIP; }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool paired = h->paired;
    const int B = b->batch, K = h->K;
    h->lastB = B;
    LcBatch bt{};
    bt.state = b->d_state; bt.action = b->d_action; bt.pi = b->d_pi_prob; bt.value = b->d_value; bt.reward = b->d_reward; bt.idx = b->d_index;
    bt.w = b->d_weights; bt.prio = b->d_priorities; bt.B = B; bt.state_i8 = b->state_is_int8; bt.action_bytes = b->action_bytes; bt.K = K; bt.A = h->A;
    bt.in_dim = h->C0 * h->hw;
    hipLaunchKernelGGL(k_lc_gather, dim3(cdiv(bt.in_dim, 256) > 8 ? 8 : cdiv(bt.in_dim, 256), B), dim3(256), 0, st, bt, h->obs, h->act);
    Sched sr{h, B, 0, false}, s0{h, B, 0, paired}, s1{h, B, 1, paired};
    // ---- forward ----
    std::vector<Op> ops, ops2;
    float* hraw = sr.tower_fwd(ops, h->tower[0], h->app_rep, h->obs, nullptr);
    for (const Op& o : ops) launch_ops(h, &o, nullptr, st);
    launch_normalize(h, hraw, h->s[0], B, st);
    std::vector<float*> g_raw(K), f_out(K);
    for (int t = 0; t < K; t++) {
        ops.clear(); ops2.clear();
        g_raw[t] = s0.tower_fwd(ops, h->tower[1], h->app_dyn[t], h->s[t], h->act + (size_t)t * B);
        f_out[t] = s1.tower_fwd(ops2, h->tower[2], h->app_pred[t], h->s[t], nullptr);
        if (run_zip(h, ops, ops2, true, st, paired)) { err = "internal: op lists of the paired towers do not line up"; return MZL_E_STATE; }
        if (t + 1 < K) launch_normalize(h, g_raw[t], h->s[t + 1], B, st);
    }
    // ---- heads ----
    const int ng = 3 * K;
    h->groups_host.resize(ng);
    for (int t = 0; t < K; t++) {
        h->groups_host[3 * t + 0] = LchGroup{g_raw[t], 0, t};
        h->groups_host[3 * t + 1] = LchGroup{f_out[t], 1, t};
        h->groups_host[3 * t + 2] = LchGroup{f_out[t], 2, t};
    }
    if (hipMemcpyAsync(h->d_groups, h->groups_host.data(), ng * sizeof(LchGroup), hipMemcpyHostToDevice, st) != hipSuccess) { err = "hipMemcpyAsync"; return MZL_E_HIP; }
    LchArgs HA{};
    for (int i = 0; i < 3; i++) { HA.head[i] = h->head[i]; HA.lwT_off[i] = h->lwT_off[i]; }
    HA.groups = h->d_groups; HA.ngroups = ng; HA.K = K; HA.B = B; HA.P = h->P; HA.hw = h->hw; HA.A = h->A;
    HA.params = h->params; HA.grads = h->grads; HA.running = h->running; HA.nbt = h->nbt; HA.lwT = h->lwT;
    HA.u = h->hu; HA.dzb = h->hdz; HA.feat = h->hfeat; HA.dlogit = h->hdl; HA.spart = h->hspart; HA.coef = h->hcoef; HA.save = h->hsave; HA.lpart = h->hlpart;
    HA.n_max = h->n_max; HA.bt = bt; HA.loss = b->d_loss; HA.wpart = h->hwpart; HA.hp_total = h->hp_total;
    for (int i = 0; i < 3; i++) HA.hp_off[i] = h->hp_off[i];
    hipLaunchKernelGGL(k_lch_conv, dim3(B, ng), dim3(256), 0, st, HA);
    hipLaunchKernelGGL(k_lch_bn, dim3(1), dim3(64 * 3 * LCH_MAXOC), 0, st, HA);
    {
        const size_t lds = ((size_t)LCH_MAXOC * h->hw + 2 * (size_t)h->n_max + 16) * sizeof(float);
        hipLaunchKernelGGL(k_lch_loss, dim3(B, ng), dim3(256), lds, st, HA);
    }
    hipLaunchKernelGGL(k_lch_bnb, dim3(1), dim3(64 * (3 * LCH_MAXOC + 1)), 0, st, HA);
    LchDx dx{};
    dx.out[0] = h->dF_pred; dx.out[1] = h->dF_rew;
    hipLaunchKernelGGL(k_lch_dx, dim3(B, K, 2), dim3(256), 0, st, HA, dx);
    hipLaunchKernelGGL(k_lch_dw1, dim3(h->P, 2, K), dim3(256), 0, st, HA);
    hipLaunchKernelGGL(k_lch_dlin, dim3(cdiv(LCH_MAXOC * h->hw, 256), cdiv(h->n_max, LCH_DLN), 3 * K), dim3(256), 0, st, HA);
    hipLaunchKernelGGL(k_lch_wsum, dim3(cdiv(h->hp_total, 256)), dim3(256), 0, st, HA);
    // ---- backward ----
    const int eg = entry_groups(h, B);
    const float* gs_next = nullptr;  // gradient wrt s_{t+1}
    float* gs_bufs[2] = {h->GsA, h->GsB};
    for (int t = K - 1; t >= 0; t--) {
        const size_t tb = (size_t)t * B * h->P * h->hw;
        LcEntry ed{};
        ed.x = g_raw[t]; ed.gs = gs_next; ed.extra = h->dF_rew + tb; ed.partner = h->app_dyn[t].y.back(); ed.dz = h->D[0][0]; ed.stat_part = h->stat[0];
        ed.scale = 0.5f; ed.B = B; ed.C = h->P; ed.hw = h->hw; ed.cpad = pad16(h->P);
        launch_entry(h, ed, st);
        LcEntry ep = ed;
        ep.x = f_out[t]; ep.gs = nullptr; ep.extra = h->dF_pred + tb; ep.partner = h->app_pred[t].y.back(); ep.dz = h->D[1][0]; ep.stat_part = h->stat[1];
        launch_entry(h, ep, st);
        float* gs_t = gs_bufs[t & 1];
        const int acc = t == K - 1 ? 0 : 1;
        ops.clear(); ops2.clear();
        s0.tower_bwd(ops, h->tower[1], h->app_dyn[t], h->s[t], h->act + (size_t)t * B, eg, acc, gs_t, h->GsP);
        s1.tower_bwd(ops2, h->tower[2], h->app_pred[t], h->s[t], nullptr, eg, acc, h->GsP, nullptr);
        if (run_zip(h, ops, ops2, false, st, paired)) { err = "internal: op lists of the paired towers do not line up"; return MZL_E_STATE; }
        gs_next = gs_t;
    }
    {
        LcEntry er{};
        er.x = hraw; er.gs = gs_next; er.extra = nullptr; er.partner = h->app_rep.y.back(); er.dz = h->D[0][0]; er.stat_part = h->stat[0];
        er.scale = 1.0f; er.B = B; er.C = h->P; er.hw = h->hw; er.cpad = pad16(h->P);
        launch_entry(h, er, st);
        ops.clear();
        sr.tower_bwd(ops, h->tower[0], h->app_rep, h->obs, nullptr, eg, 0, nullptr, nullptr);
        for (const Op& o : ops) launch_ops(h, &o, nullptr, st);
    }
    if (hipGetLastError() != hipSuccess) { err = "a conv-learner kernel failed to launch"; return MZL_E_HIP; }
    return MZL_OK;
}

int mzlc_apply(mzlc_learner* h, double lr, double beta1, double beta2, double eps, double weight_decay, double max_grad_norm, int64_t step, void* stream,
               std::string& err) {
    if (!h->committed) { err = "weights not committed"; return MZL_E_STATE; }
    if (hipSetDevice(h->device) != hipSuccess) { err = "hipSetDevice"; return MZL_E_HIP; }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool clip = max_grad_norm > 0.0;
    if (clip) hipLaunchKernelGGL(k_lc_sqsum, dim3(h->sq_blocks), dim3(256), 0, st, h->grads, (int)h->total, h->d_sq);
    LcAdam a{};
    a.lr = (float)lr; a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps; a.weight_decay = (float)weight_decay;
    a.max_norm = clip ? (float)max_grad_norm : 0.0f;
    a.bc1 = (float)(1.0 - std::pow(beta1, (double)step));
    a.bc2 = (float)(1.0 - std::pow(beta2, (double)step));
    a.sq_blocks = h->sq_blocks; a.n = (int)h->total;
    hipLaunchKernelGGL(k_lc_adam, dim3(cdiv((int)h->total, 256)), dim3(256), 0, st, h->params, h->grads, h->m, h->v, h->d_sq, a);
    if (pack_all(h, st) != MZL_OK) { err = "pack kernels failed to launch"; return MZL_E_HIP; }
    return MZL_OK;
}

// diagnostic (tests): device pointers of saved tensors.  what: "y" (a = application: 0 representation, 1 + t dynamics_t, 1 + K + t prediction_t;
// b = layer of the tower), "x" (materialised outputs), "s" (a = t), "dF_pred" / "dF_rew" (a = t)
int mzlc_debug_tensor(const mzlc_learner* h, const char* what, int a, int b, void** ptr, int64_t* count) {
    const std::string w = what;
    const AppBufs* ap = nullptr;
    if (a == 0) ap = &h->app_rep;
    else if (a >= 1 && a <= h->K) ap = &h->app_dyn[a - 1];
    else if (a > h->K && a <= 2 * h->K) ap = &h->app_pred[a - 1 - h->K];
    *count = (int64_t)h->T;
    if (w == "y" && ap && b >= 0 && b < (int)ap->y.size()) { *ptr = ap->y[b]; return MZL_OK; }
    if (w == "x" && ap && b >= 0 && b < (int)ap->x.size()) { *ptr = ap->x[b]; return MZL_OK; }
    if (w == "s" && a >= 0 && a < h->K) { *ptr = h->s[a]; return MZL_OK; }
    const size_t tb = (size_t)h->lastB * h->P * h->hw;
    if (w == "dF_pred" && a >= 0 && a < h->K) { *ptr = h->dF_pred + (size_t)a * tb; return MZL_OK; }
    if (w == "dF_rew" && a >= 0 && a < h->K) { *ptr = h->dF_rew + (size_t)a * tb; return MZL_OK; }
    if (w == "gs") { *ptr = a ? h->GsB : h->GsA; return MZL_OK; }
    return MZL_E_INVALID;
}
