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
    int R = 0;                // residual blocks
};
// launch geometry of the whole-image conv / weight-gradient kernels for h x w "images" (boards, hidden states, tiles with their halo)
struct Geom {
    int h = 0, w = 0, hw = 0;
    int npt = 15, G = 1, qstride = 0;       // k_lc_conv: pixel tiles per workgroup, images per workgroup, LDS slot-plane stride
    bool side15 = false;                     // the 15 x 15 build
    bool stack_rows = false;
    int P4 = 0, nsteps = 0, SPY = 0, SPX = 0, SG = 1;  // k_lc_wgrad: row pitch, 16-position steps per staging round, plane strides, images per round
};
struct TensorInfo {
    std::string name;
    int64_t off;
    int rows, cols;
};
struct AppBufs {  // saved tensors of ONE application of a tower (representation: 1, dynamics / prediction: K)
    std::vector<float*> y, x, fcoef, save, bcoef;
    std::vector<float*> dz;  // defer_wgrad: dz of every layer's BatchNorm output, kept until the steps' weight gradients run in one launch per layer
};
enum OpKind { OP_CONV, OP_WGRAD, OP_WREDUCE, OP_BNFWD, OP_BNBWD, OP_APPLY, OP_WGRAD_ACT };
struct Op {
    int kind;
    int npt = 15, side15 = 0;  // OP_CONV: which build
    LcConv conv;
    LcWgrad wg;
    LcWreduce wr;
    LcWgradAct wa;
    LcBnFwd bf;
    LcBnBwd bb;
    LcApply ap;
};

constexpr int ACT_CHUNKS = 8;  // batch chunks of k_lc_wgrad_act
int pad16(int x) { return (x + 15) & ~15; }
int cdiv(int a, int b) { return (a + b - 1) / b; }

// pixel tiling of the conv kernels (G whole images per workgroup in NPT tiles of 16 pixel slots; lane = pixel quad while staging) and the pitch
// layout of the weight-gradient kernel for h x w images; false if the image does not fit.  `images` > 0: the batch the learner is built for --
// the tiling is then chosen by the time of one launch, (rounds of workgroups over the CUs) x (tiles per workgroup), `wg_per_group` workgroups
// per image group (output-channel blocks x paired jobs): at batch 128 a 6 x 6 hidden state packs best as 4 images in 9 tiles, but that is 128
// workgroups on 256 CUs, and 2 images in 6 tiles fills the chip in 2/3 of the time.  Ties / images == 0: the densest packing.
bool make_geom(Geom& g, int hh, int ww, bool allow_side15, int images = 0, int wg_per_group = 1, int cus = 256, bool stack_wgrad = true, bool allow16 = false,
               int wgrad_blocks = 0) {
    g.h = hh; g.w = ww; g.hw = hh * ww;
    if (g.hw < 1 || g.hw > (allow16 ? 256 : 240)) return false;
    const int QP = (g.hw + 3) / 4;
    double best = -1.0;
    long best_cost = -1;
    const int cand[6] = {5, 6, 9, 13, 15, 16};  // (5: two 6 x 6 images in 80 slots; 13: a 14 x 14 image -- a 12 x 12 tile with its halo -- in 208 slots;
                                               //  16: the wide tiles of the Atari net's 48 x 48 stage only -- 14 x 18 positions)
    for (int i = 0; i < (allow16 ? 6 : 5); i++) {
        int G = (16 * cand[i]) / g.hw;
        if (64 / QP < G) G = 64 / QP;
        if (G < 1) continue;
        const double eff = (double)G * g.hw / (16.0 * cand[i]);
        const long cost = images > 0 ? (long)cdiv(cdiv(images, G) * wg_per_group, cus) * cand[i] : 0;
        // (ties: the denser packing, then the smaller tiling -- fewer accumulators per wave)
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && eff > best + 1e-9)) { best = eff; best_cost = cost; g.npt = cand[i]; g.G = G; }
    }
    if (best < 0.0) return false;
    g.qstride = (4 * g.G * (g.h + 2) * (g.w + 2) + 63) & ~63;
    g.side15 = allow_side15 && g.h == 15 && g.w == 15 && g.G == 1 && g.npt == 15;
    g.P4 = 4 * cdiv(g.w + 1, 4);
    // k_lc_wgrad stages SG images per round, SIDE BY SIDE in one wide "image" of SG (w + 1) columns (the zero column behind every image's rows is the
    // pitch layout's own row separator; the zero rows above and below are shared): a 6 x 6 plane is 9 pixel quads -- 9 of a wave's 64 staging lanes and
    // two barriers per image otherwise.  As many as the staging lanes hold and as leave two workgroups per CU their LDS (80 KB each).
    // MZLC_STACK_ROWS=1 at create: the first form of this -- images stacked vertically with a zero row between them (more padding: kept for A/B)
    // Among the counts that fit, the one with the shortest launch at the batch the learner is built for (round 6): workgroup rounds over the CUs' two slots x
    // staging rounds per workgroup x steps per round, with wgrad_ops' own chunking -- a 6 x 6 plane takes 5 per round, but at batch 128 that is 13 chunks
    // of 10 images = 416 workgroups of a paired launch on 512 slots; 4 per round is 16 chunks of 8 = 512, and 11 steps for 14 (Atari update: 29.0 -> 28.55 ms).
    g.SG = 1;
    g.stack_rows = getenv("MZLC_STACK_ROWS") != nullptr;
    const int sg_cap = getenv("MZLC_WGRAD_SG") ? atoi(getenv("MZLC_WGRAD_SG")) : 16;
    if (stack_wgrad) {
        long best_c = 0;
        bool have = false;
        for (int sg = 1; sg <= sg_cap && sg * QP <= 64; sg++) {
            const int p4 = (g.stack_rows || sg == 1) ? g.P4 : 4 * cdiv(sg * (g.w + 1), 4);
            const int ns = (g.stack_rows && sg > 1) ? cdiv((sg * (g.h + 1) - 1) * p4, 16) : cdiv(g.h * p4, 16);
            if (sg > 1 && ((size_t)32 * (32 * ns + 2 * p4 + 16) + 160) * sizeof(float) > 80 * 1024) continue;
            long c = -(long)sg;  // (no batch given: as many as fit)
            if (images > 0 && wgrad_blocks > 0 && !getenv("MZLC_WGRAD_SG_MAX")) {
                int chunks = cus / wgrad_blocks;  // (the towers' layers run paired: wgrad_ops)
                chunks = chunks < 1 ? 1 : (chunks > images ? images : chunks);
                const int ipw = cdiv(cdiv(images, chunks), sg) * sg;
                c = (long)cdiv(cdiv(images, ipw) * wgrad_blocks * 2, 2 * cus) * (ipw / sg) * ns;
            }
            if (!have || c <= best_c) { best_c = c; g.SG = sg; have = true; }  // (ties: more images per round)
        }
    }
    if (g.SG > 1 && !g.stack_rows) g.P4 = 4 * cdiv(g.SG * (g.w + 1), 4);
    g.nsteps = (g.SG > 1 && g.stack_rows) ? cdiv((g.SG * (g.h + 1) - 1) * g.P4, 16) : cdiv(g.h * g.P4, 16);
    g.SPY = 16 * g.nsteps + 4;
    g.SPX = 2 * g.P4 + 16 * g.nsteps + 12;
    return true;
}

}  // namespace

struct mzlc_learner {
    mzl_config cfg{};
    int device = 0, num_cus = 256;
    int P = 0, C0 = 0, A = 0, R = 0, K = 0, h = 0, w = 0, hw = 0, maxB = 0;
    Geom gm;                 // geometry of the hidden state (the board; 6 x 6 for the Atari net)
    int wgrad_min_ipw = 1;   // k_lc_wgrad: least images per workgroup (small images: the partial tile's write-out amortised over more of them)
    bool xcd_remap = true;   // k_lc_wgrad: the blocks of one image chunk on one XCD (MZLC_NO_XCD_REMAP=1 at create: launch order)
    bool fuse_apply = true;  // block outputs formed in the next conv's staging (MZLC_NO_FUSE_APPLY=1 at create: one k_lc_apply per block)
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
    float* wpart_act = nullptr;  // k_lc_wgrad_act: [ACT_CHUNKS][P][A][9]
    bool act_sparse = true;      // the action planes' weight gradient as a gather (MZLC_ACT_MFMA=1 at create: inside the MFMA kernel, the first form)
    float* D[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    float *GsA = nullptr, *GsB = nullptr, *GsP = nullptr;
    float *dF_pred = nullptr, *dF_rew = nullptr;
    // heads
    LchGroup* d_groups = nullptr;
    std::vector<LchGroup> groups_host, groups_dev;  // this call's table; the one d_groups holds
    float *hu = nullptr, *hdz = nullptr, *hfeat = nullptr, *hdl = nullptr, *hspart = nullptr, *hspiv = nullptr, *hcoef = nullptr, *hsave = nullptr, *hlpart = nullptr, *hwpart = nullptr;
    int hp_off[3] = {0, 0, 0}, hp_total = 0;
    int n_max = 1;
    float* d_sq = nullptr;
    int sq_blocks = 0;
    const void* checked_ptr[9] = {};
    // ---- MuZeroAtariNet (net_kind == MZL_NET_ATARI): the representation net works on 96 x 96 -> 48 x 48 -> 24 x 24 planes before the hidden state's 6 x 6 ----
    bool atari = false;
    int obsH = 0, obsW = 0;
    Geom gt, g12;                 // 14 x 14 tiles (12 x 12 + halo) of the large planes; the 12 x 12 stage as whole images
    int l_c1 = -1, l_c2 = -1;     // conv_1 / conv_2 (stride 2, no BatchNorm)
    std::vector<int> l_b1, l_b2;  // the 4 convs of res_blocks_1 (48 x 48, 128 planes) / res_blocks_2 (24 x 24, num_planes)
    TowerInfo t12;                // res_blocks_3 (12 x 12): the fused whole-image path
    AppBufs a12;
    struct StageBufs { std::vector<float*> y, h1, x, fcoef, save, bcoef, xt; float *dzA = nullptr, *dzB = nullptr, *gF = nullptr; };  // xt: the forward convs' gathered input tiles, kept for the weight gradient (round 6)
    StageBufs sb48, sb24;
    float *y_c1 = nullptr, *a1 = nullptr, *y_c2 = nullptr, *a2 = nullptr, *p1 = nullptr, *hraw = nullptr, *g12in = nullptr, *dH = nullptr;
    float *TA = nullptr, *TB = nullptr, *TC = nullptr;      // tile buffers
    float* D12[3] = {nullptr, nullptr, nullptr};            // the gradient ping-pong buffers of the 12 x 12 stage
    float *coef_relu = nullptr, *coef_ident = nullptr;      // [3][cpad]: (a, b) = (1, 0); (c1, c2, c3) = (1, 0, 0)
    int par_f[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, par_d[4] = {0, 0, 0, 0};  // packed offsets of the parity copies: forward of conv_1 / conv_2, data gradient of conv_2
    LcPackPar* d_pack_par = nullptr;
    int n_pack_par = 0;
    Geom gt16;                // 12 x 16 tiles with their halo (14 x 18): planes whose width 16 divides into at least three (the 48 x 48 stage)
    bool wide_tiles = true;   // MZLC_NO_WIDE_TILES=1 at create: 12 x 12 tiles everywhere
    bool bad_dispatch = false;  // a launch found no kernel build for its job (reported by mzlc_grad: never silently skipped)
    bool par_compact = true;  // the parity planes of conv_1 / conv_2 on their own taps only (MZLC_NO_TAPSETS=1 at create: nine taps, zero weights)
    bool halo_in = true;      // the tiled stages' stride-1 convs compute the inner positions of a tile only (MZLC_NO_HALO_IN=1 at create: whole haloed tiles)
    bool ring_rows = true;    // the tiled stages' weight gradients reduce over the inner rows of a tile only (MZLC_NO_RING_ROWS=1 at create: all rows, ring zeroed)
    bool defer_wgrad = false; // the shared towers' block layers: weight gradients of all K unroll steps in one launch per layer, after the last step's backward
                              // (decided at create: where one step's batch is <= 4 staging rounds per workgroup -- small planes; MZLC_DEFER_WGRAD=0 / 1 overrides)
    LcWgradSrc* d_srcs = nullptr;  // [tower 1, 2][layer][K]
    int srcs_stride[2] = {0, 0};   // layers of the tower
    bool fuse_entry = true;   // (with out_plane) the tiled stages' data-gradient convs mask their result and take the BatchNorm-backward sums themselves
                              // (MZLC_NO_FUSE_ENTRY=1: a gradient plane + k_lc_entry_plain)
    bool skip_h1 = true;      // (with fuse_entry and keep_tiles) the tiled stages' inner activation relu(bn(conv1)) exists as TILES only -- formed by the gather from
                              // the raw conv output -- and the backward mask is the sign of a y + b as in the towers (MZLC_KEEP_H1=1: the plane too; same bits)
    bool quad_steps = true;   // (with ring_rows) 12-wide tiles: a reduction step is four real quads in row order, the rows' pad quads are skipped (MZLC_NO_QUAD_STEPS=1)
    bool row_steps = true;    // (with ring_rows) a reduction step of those weight gradients is one row of a wide tile's 16 inner columns (MZLC_NO_ROW_STEPS=1: 16 flat positions)
    bool out_plane = true;    // (with halo_in) the tiled stages' stride-1 convs write their outputs straight into the plane, add the skip from there and sum the
                              // BatchNorm statistics per tile (MZLC_NO_OUT_PLANE=1 at create: inner-only tiles + k_lc_tile_scatter)
    bool keep_tiles = true;   // the forward pass's gathered input tiles stay in HBM (0.8 GB at batch 128) and ARE the weight gradient's x operand
                              // (MZLC_NO_KEEP_TILES=1 at create: gathered again in the backward pass, round 5's form; same bits)
    int max_imgs = 0;             // images a conv / weight-gradient launch may see (batch x 16 tiles for the Atari net)
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
    t.R = h->R;
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

bool alloc_app(mzlc_learner* h, const TowerInfo& t, AppBufs& a, size_t elems, bool with_dz = false) {
    bool ok = true;
    const int nl = (int)t.layers.size(), nx = t.R + (t.conv0 ? 1 : 0);
    if (with_dz) {
        a.dz.resize(nl);
        for (int i = 0; i < nl; i++) ok = ok && dalloc(h, &a.dz[i], elems) == hipSuccess;
    }
    a.y.resize(nl); a.fcoef.resize(nl); a.save.resize(nl); a.bcoef.resize(nl); a.x.resize(nx);
    const int cpad = pad16(h->P > 128 ? h->P : 128);
    for (int i = 0; i < nl; i++) {
        ok = ok && dalloc(h, &a.y[i], elems) == hipSuccess && dalloc(h, &a.fcoef[i], (size_t)3 * cpad) == hipSuccess &&
             dalloc(h, &a.save[i], (size_t)2 * cpad) == hipSuccess && dalloc(h, &a.bcoef[i], (size_t)3 * cpad) == hipSuccess;
    }
    for (int i = 0; i < nx; i++) ok = ok && dalloc(h, &a.x[i], elems) == hipSuccess;
    return ok;
}

size_t conv_lds(int qstride, int cpad_in) { return ((size_t)8 * qstride + (size_t)3 * cpad_in) * sizeof(float); }
size_t wgrad_lds(const Geom& g) { return ((size_t)32 * (g.SPY + g.SPX) + 160) * sizeof(float); }

// ---- op builders -------------------------------------------------------------------------------------------------------------
struct Sched {
    mzlc_learner* h;
    int B, lane;      // B: images of this geometry (batch, or batch x tiles)
    bool lane_pairs;  // this tower's launches are paired with another tower's
    Geom g;           // geometry of the images these ops work on
    int C;            // channels of the activation tensors (num_planes; 128 in the Atari net's first stage)
    int groups() const { return cdiv(B, g.G); }

    LcConv conv_base(const LayerInfo& L, bool dgrad) const {
        LcConv c{};
        c.B = B; c.G = g.G; c.h = g.h; c.w_img = g.w; c.qstride = g.qstride;
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
    Op op_conv(const LcConv& c) const { Op o{}; o.kind = OP_CONV; o.conv = c; o.npt = g.npt; o.side15 = g.side15 ? 1 : 0; return o; }
    Op op_bnfwd(const LayerInfo& L, float* fcoef, float* save) const {
        Op o{};
        o.kind = OP_BNFWD;
        LcBnFwd& f = o.bf;
        f.part = h->stat[lane]; f.gamma = h->params + L.bn.gamma_off; f.beta = h->params + L.bn.beta_off; f.coef = fcoef; f.save = save;
        f.running_mean = h->running ? h->running + L.bn.rm_off : nullptr; f.running_var = h->running ? h->running + L.bn.rv_off : nullptr;
        f.num_batches = h->nbt ? h->nbt + L.bn.nbt : nullptr;
        f.groups = groups(); f.C = L.bn.C; f.cpad = L.bn.cpad; f.count = (float)B * (float)g.hw;
        return o;
    }
    Op op_bnbwd(const LayerInfo& L, const float* save, float* bcoef, int ngroups, int accumulate) const {
        Op o{};
        o.kind = OP_BNBWD;
        LcBnBwd& f = o.bb;
        f.part = h->stat[lane]; f.gamma = h->params + L.bn.gamma_off; f.save = save; f.coef = bcoef;
        f.dgamma = h->grads + L.bn.gamma_off; f.dbeta = h->grads + L.bn.beta_off;
        f.groups = ngroups; f.C = L.bn.C; f.cpad = L.bn.cpad; f.accumulate = accumulate; f.count = (float)B * (float)g.hw;
        return o;
    }
    Op op_apply(const float* y, const float* res, const float* coef, float* out) const {
        Op o{};
        o.kind = OP_APPLY;
        o.ap.y = y; o.ap.res = res; o.ap.coef = coef; o.ap.out = out; o.ap.C = C; o.ap.hw = g.hw; o.ap.cpad = pad16(C);
        o.ap.n = (long long)B * C * g.hw;
        return o;
    }
    void wgrad_ops(std::vector<Op>& ops, const LayerInfo& L, const float* dz, const float* y, const float* bcoef, const float* x0, int x_mode, const float* xcoef,
                   const int* action, int accumulate) const {
        // hidden channels here, action planes by k_lc_wgrad_act -- where that removes at least four 16-channel tiles from the MFMA work (board games:
        // A = hw + 1 planes; the Atari net's 6-18 planes ride in the hidden channels' last tile and stay)
        const bool split = action && h->act_sparse && cdiv(L.cin, 16) - cdiv(L.cin_real, 16) >= 4;
        Op o{};
        o.kind = OP_WGRAD;
        LcWgrad& g = o.wg;
        g.dz = dz; g.y = y; g.dcoef = bcoef; g.x0 = x0; g.xcoef = xcoef; g.x_mode = x_mode; g.action = split ? nullptr : action; g.num_actions = h->A;
        g.cin_real = L.cin_real; g.cin = split ? L.cin_real : L.cin; g.cout = L.cout; g.ci_tiles = cdiv(g.cin, 16); g.co_tiles = L.co_tiles;
        g.cpad_in = pad16(L.cin_real); g.cpad_out = pad16(L.cout);
        g.B = B; g.h = this->g.h; g.w_img = this->g.w; g.P4 = this->g.P4; g.nsteps = this->g.nsteps; g.SPY = this->g.SPY; g.SPX = this->g.SPX; g.sg = this->g.SG; g.sg_cols = (this->g.SG > 1 && !this->g.stack_rows) ? 1 : 0;
        g.co_blocks = cdiv(g.co_tiles, 2);
        const int ci_blocks = cdiv(g.ci_tiles, 2);
        // two workgroups per CU in all: a paired launch brings the other half; the first conv blocks (action planes: the dynamics tower's extra
        // ops) are never paired
        int chunks = ((lane_pairs && !action) ? 1 : 2) * h->num_cus / (g.co_blocks * ci_blocks);
        chunks = chunks < 1 ? 1 : (chunks > B ? B : chunks);
        g.ipw = cdiv(B, chunks);
        if (g.ipw < h->wgrad_min_ipw) g.ipw = h->wgrad_min_ipw < B ? h->wgrad_min_ipw : B;
        g.ipw = cdiv(g.ipw, g.sg) * g.sg;  // whole staging rounds
        chunks = cdiv(B, g.ipw);
        g.part = h->wpart[lane];
        ops.push_back(o);
        Op r{};
        r.kind = OP_WREDUCE;
        r.wr.part = g.part; r.wr.grad = h->grads + L.w_off; r.wr.chunks = chunks; r.wr.cout = L.cout; r.wr.cin = L.cin;
        r.wr.co_pad = g.co_tiles * 16; r.wr.ci_pad = g.ci_tiles * 16; r.wr.accumulate = accumulate;
        r.wr.cin_loop = split ? L.cin_real : 0;
        ops.push_back(r);
        if (split) {
            Op a{};
            a.kind = OP_WGRAD_ACT;
            LcWgradAct& w = a.wa;
            w.dz = dz; w.y = y; w.dcoef = bcoef; w.action = action; w.part = h->wpart_act; w.grad = h->grads + L.w_off;
            w.B = B; w.cout = L.cout; w.cpad_out = pad16(L.cout); w.cin = L.cin; w.cin_real = L.cin_real; w.A = h->A; w.h = this->g.h; w.w = this->g.w;
            w.nchunk = B < ACT_CHUNKS ? B : ACT_CHUNKS; w.bchunk = cdiv(B, w.nchunk); w.nchunk = cdiv(B, w.bchunk); w.accumulate = accumulate;
            ops.push_back(a);
        }
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
        for (int r = 0; r < t.R; r++) {
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
            if (h->fuse_apply && r + 1 < t.R) {
                pend_y = a.y[l2]; pend_coef = a.fcoef[l2]; pend_res = cur; pend_out = a.x[xi + r];
            } else {
                ops.push_back(op_apply(a.y[l2], cur, a.fcoef[l2], a.x[xi + r]));
                cur = a.x[xi + r];
            }
        }
        return const_cast<float*>(cur);
    }

    // weight gradient of a block layer the K unroll steps share, all steps in one launch (LcWgrad::srcs); x_mode: IN_BNRELU (second conv of a block)
    // or IN_IDENT.  Chunks: as many per step as keep the launch at two workgroups per CU (one with a paired launch), at least one.
    void wgrad_ops_steps(std::vector<Op>& ops, const LayerInfo& L, const LcWgradSrc* srcs, int nsrc, int x_mode) const {
        Op o{};
        o.kind = OP_WGRAD;
        LcWgrad& g = o.wg;
        g.srcs = srcs; g.nsrc = nsrc; g.x_mode = x_mode; g.num_actions = h->A;
        g.cin_real = L.cin_real; g.cin = L.cin; g.cout = L.cout; g.ci_tiles = cdiv(g.cin, 16); g.co_tiles = L.co_tiles;
        g.cpad_in = pad16(L.cin_real); g.cpad_out = pad16(L.cout);
        g.B = B; g.h = this->g.h; g.w_img = this->g.w; g.P4 = this->g.P4; g.nsteps = this->g.nsteps; g.SPY = this->g.SPY; g.SPX = this->g.SPX; g.sg = this->g.SG; g.sg_cols = (this->g.SG > 1 && !this->g.stack_rows) ? 1 : 0;
        g.co_blocks = cdiv(g.co_tiles, 2);
        const int ci_blocks = cdiv(g.ci_tiles, 2);
        int cps = (lane_pairs ? 1 : 2) * h->num_cus / (g.co_blocks * ci_blocks) / nsrc;
        cps = cps < 1 ? 1 : (cps > B ? B : cps);
        g.ipw = cdiv(B, cps);
        if (g.ipw < h->wgrad_min_ipw) g.ipw = h->wgrad_min_ipw < B ? h->wgrad_min_ipw : B;
        g.ipw = cdiv(g.ipw, g.sg) * g.sg;  // whole staging rounds
        g.cps = cdiv(B, g.ipw);
        g.part = h->wpart[lane];
        ops.push_back(o);
        Op r{};
        r.kind = OP_WREDUCE;
        r.wr.part = g.part; r.wr.grad = h->grads + L.w_off; r.wr.chunks = nsrc * g.cps; r.wr.cout = L.cout; r.wr.cin = L.cin;
        r.wr.co_pad = g.co_tiles * 16; r.wr.ci_pad = g.ci_tiles * 16; r.wr.accumulate = 0;
        ops.push_back(r);
    }

    // backward of one tower application.  D[lane][0] holds dz of the tower's last BatchNorm, its partial sums are in stat[lane] (entry_groups groups).
    // final_out: where the gradient wrt the tower's input goes (null: not needed -- the representation tower); final_skip: added to it.
    void tower_bwd(std::vector<Op>& ops, const TowerInfo& t, AppBufs& a, const float* x_in, const int* action, int entry_groups, int accumulate,
                   float* final_out, const float* final_skip, bool defer = false) const {
        // defer (h->defer_wgrad, the shared towers): every layer's dz goes to its own buffer of the application (a.dz) and the block layers' weight
        // gradients are left to wgrad_ops_steps, after the last step's backward; the caller's entry kernel wrote a.dz.back()
        float *Da = defer ? a.dz.back() : h->D[lane][0], *Db = h->D[lane][1], *Dc = h->D[lane][2];
        const int li = t.conv0 ? 1 : 0, xi = t.conv0 ? 1 : 0;
        int ng = entry_groups;
        for (int r = t.R - 1; r >= 0; r--) {
            const int l1 = li + 2 * r, l2 = l1 + 1;
            if (defer) { Db = a.dz[l1]; Dc = l1 > 0 ? a.dz[l1 - 1] : h->D[lane][2]; }
            const LayerInfo &L1 = h->layers[t.layers[l1]], &L2 = h->layers[t.layers[l2]];
            const float* xin_blk = r > 0 ? a.x[xi + r - 1] : (t.conv0 ? a.x[0] : x_in);
            ops.push_back(op_bnbwd(L2, a.save[l2], a.bcoef[l2], ng, accumulate));
            if (!defer) wgrad_ops(ops, L2, Da, a.y[l2], a.bcoef[l2], a.y[l1], IN_BNRELU, a.fcoef[l1], nullptr, accumulate);
            LcConv c = conv_base(L2, true);
            c.in0 = Da; c.in1 = a.y[l2]; c.coef = a.bcoef[l2]; c.in_mode = IN_BNBWD; c.out = Db;
            c.mask = a.y[l1]; c.mcoef = a.fcoef[l1]; c.partner = a.y[l1]; c.stat_mode = ST_BWD; c.stat_part = h->stat[lane];
            ops.push_back(op_conv(c));
            ng = groups();
            ops.push_back(op_bnbwd(L1, a.save[l1], a.bcoef[l1], ng, accumulate));
            if (!defer) wgrad_ops(ops, L1, Db, a.y[l1], a.bcoef[l1], xin_blk, IN_IDENT, nullptr, nullptr, accumulate);
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
// the tap sets of the parity planes (par_tapmap): forward rows {1} | {0, 1}, data-gradient rows {1} | {1, 2}, squared
template <int NPT, int MASK>
void launch_conv_taps(const Pair<LcConv>& pj, dim3 grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_lc_conv<NPT, IN_IDENT, 0, MASK>), grid, dim3(256), lds, st, pj);
}
// (the data-gradient planes exist for conv_2 only: 24 x 24 planes, 12 x 12 tiles, NPT 15)
bool launch_conv_tapmask(int npt, int mask, const Pair<LcConv>& pj, dim3 grid, size_t lds, hipStream_t st) {
    if (npt == 16) {
        switch (mask) {
            case 0x010: launch_conv_taps<16, 0x010>(pj, grid, lds, st); return true;
            case 0x018: launch_conv_taps<16, 0x018>(pj, grid, lds, st); return true;
            case 0x012: launch_conv_taps<16, 0x012>(pj, grid, lds, st); return true;
            case 0x01b: launch_conv_taps<16, 0x01b>(pj, grid, lds, st); return true;
        }
        return false;
    }
    if (npt == 13) {
        switch (mask) {
            case 0x010: launch_conv_taps<13, 0x010>(pj, grid, lds, st); return true;
            case 0x018: launch_conv_taps<13, 0x018>(pj, grid, lds, st); return true;
            case 0x012: launch_conv_taps<13, 0x012>(pj, grid, lds, st); return true;
            case 0x01b: launch_conv_taps<13, 0x01b>(pj, grid, lds, st); return true;
            case 0x030: launch_conv_taps<13, 0x030>(pj, grid, lds, st); return true;
            case 0x090: launch_conv_taps<13, 0x090>(pj, grid, lds, st); return true;
            case 0x1b0: launch_conv_taps<13, 0x1b0>(pj, grid, lds, st); return true;
        }
        return false;
    }
    if (npt != 15) return false;
    switch (mask) {
        case 0x010: launch_conv_taps<15, 0x010>(pj, grid, lds, st); return true;
        case 0x018: launch_conv_taps<15, 0x018>(pj, grid, lds, st); return true;
        case 0x012: launch_conv_taps<15, 0x012>(pj, grid, lds, st); return true;
        case 0x01b: launch_conv_taps<15, 0x01b>(pj, grid, lds, st); return true;
        case 0x030: launch_conv_taps<15, 0x030>(pj, grid, lds, st); return true;
        case 0x090: launch_conv_taps<15, 0x090>(pj, grid, lds, st); return true;
        case 0x1b0: launch_conv_taps<15, 0x1b0>(pj, grid, lds, st); return true;
    }
    return false;
}
template <int NPT, int MASK>
hipError_t conv_taps_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_conv<NPT, IN_IDENT, 0, MASK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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
            if (b && (b->conv.in_mode != a->conv.in_mode || b->npt != a->npt || b->side15 != a->side15)) {  // (never the case for the zipped towers; kept correct anyway)
                launch_ops(h, a, nullptr, st);
                return launch_ops(h, b, nullptr, st);
            }
            // (the forward statistics are taken from the accumulators themselves: a conv that leaves them carries no skip and no mask)
            if ((a->conv.stat_mode == ST_FWD && (a->conv.skip || a->conv.mask)) || (b && b->conv.stat_mode == ST_FWD && (b->conv.skip || b->conv.mask))) {
                h->bad_dispatch = true;
                return MZL_E_INVALID;
            }
            Pair<LcConv> pj{};
            pj.a = a->conv;
            const int ga = cdiv(a->conv.B, a->conv.G), gb = b ? cdiv(b->conv.B, b->conv.G) : 0;
            pj.na = ga;
            if (b) pj.b = b->conv;
            int z = cdiv(a->conv.co_tiles, 4), cp = a->conv.cpad_in;
            if (b) { z = cdiv(b->conv.co_tiles, 4) > z ? cdiv(b->conv.co_tiles, 4) : z; cp = b->conv.cpad_in > cp ? b->conv.cpad_in : cp; }
            const dim3 grid(1, ga + gb, z);
            const size_t lds = conv_lds(a->conv.qstride, cp);
            const int mode = a->conv.in_mode;
            if (a->conv.tapmask && a->conv.tapmask != 0x1ff) {  // (a parity plane: built as whole-tile, identity-mode, unpaired launches only)
                if (b || a->side15 || mode != IN_IDENT || !launch_conv_tapmask(a->npt, a->conv.tapmask, pj, grid, lds, st)) { h->bad_dispatch = true; return MZL_E_INVALID; }
            } else
            if (a->conv.out_plane) {  // the tile path's stride-1 convs writing planes (halo_in; 12 x 12 or 12 x 16 inner positions)
                if (mode != IN_IDENT || b || !a->conv.halo_in || (a->npt != 9 && a->npt != 12)) { h->bad_dispatch = true; return MZL_E_INVALID; }
                if (a->npt == 12) hipLaunchKernelGGL((k_lc_conv<12, IN_IDENT, 0, 0x1ff, true>), grid, dim3(256), lds, st, pj);
                else hipLaunchKernelGGL((k_lc_conv<9, IN_IDENT, 0, 0x1ff, true>), grid, dim3(256), lds, st, pj);
            } else
            if (a->npt == 16) {  // the wide tiles of the Atari net (gathered: identity staging)
                if (mode != IN_IDENT) { h->bad_dispatch = true; return MZL_E_INVALID; }
                hipLaunchKernelGGL((k_lc_conv<16, IN_IDENT, 0>), grid, dim3(256), lds, st, pj);
            } else
            if (a->npt == 12) {  // the inner 12 x 16 of a wide tile (halo_in)
                if (mode != IN_IDENT || b || !a->conv.halo_in) { h->bad_dispatch = true; return MZL_E_INVALID; }
                hipLaunchKernelGGL((k_lc_conv<12, IN_IDENT, 0>), grid, dim3(256), lds, st, pj);
            } else
            if (a->npt == 15 && a->side15) launch_conv<15, 15>(mode, pj, grid, lds, st);
            else if (a->npt == 15) launch_conv<15, 0>(mode, pj, grid, lds, st);
            else if (a->npt == 13) launch_conv<13, 0>(mode, pj, grid, lds, st);
            else if (a->npt == 9) launch_conv<9, 0>(mode, pj, grid, lds, st);
            else if (a->npt == 6) launch_conv<6, 0>(mode, pj, grid, lds, st);
            else launch_conv<5, 0>(mode, pj, grid, lds, st);
            break;
        }
        case OP_WGRAD: {
            Pair<LcWgrad> pj{};
            pj.a = a->wg;
            auto wg_chunks = [](const LcWgrad& g) { return (g.srcs ? g.nsrc : 1) * cdiv(g.B, g.ipw); };
            const int ya = a->wg.co_blocks * wg_chunks(a->wg), yb = b ? b->wg.co_blocks * wg_chunks(b->wg) : 0;
            pj.na = ya;
            if (b) pj.b = b->wg;
            int x = cdiv(a->wg.ci_tiles, 2);
            if (b && cdiv(b->wg.ci_tiles, 2) > x) x = cdiv(b->wg.ci_tiles, 2);
            // a chunk's blocks on one XCD (k_lc_wgrad): needs the same block grid in both jobs and a group count the 8 XCDs divide
            const int groups = (ya + yb) / a->wg.co_blocks;
            pj.remap = (h->xcd_remap && (!b || (b->wg.co_blocks == a->wg.co_blocks && cdiv(b->wg.ci_tiles, 2) == cdiv(a->wg.ci_tiles, 2))) && groups % 8 == 0) ? 1 : 0;
            const size_t wlds = ((size_t)32 * (a->wg.SPY + a->wg.SPX) + 160) * sizeof(float);
            const dim3 wgrid(x, ya + yb);
            if (a->wg.tapmask && a->wg.tapmask != 0x1ff) {  // a parity plane (tile path: ring_zero, no action planes, unpaired)
                if (b || !a->wg.ring_zero || a->wg.action) { h->bad_dispatch = true; return MZL_E_INVALID; }
                switch (a->wg.tapmask) {
                    case 0x010: hipLaunchKernelGGL((k_lc_wgrad<false, true, 0x010>), wgrid, dim3(256), wlds, st, pj); break;
                    case 0x018: hipLaunchKernelGGL((k_lc_wgrad<false, true, 0x018>), wgrid, dim3(256), wlds, st, pj); break;
                    case 0x012: hipLaunchKernelGGL((k_lc_wgrad<false, true, 0x012>), wgrid, dim3(256), wlds, st, pj); break;
                    case 0x01b: hipLaunchKernelGGL((k_lc_wgrad<false, true, 0x01b>), wgrid, dim3(256), wlds, st, pj); break;
                    default: h->bad_dispatch = true; return MZL_E_INVALID;
                }
            } else
            if (a->wg.ring_zero) hipLaunchKernelGGL((k_lc_wgrad<false, true>), dim3(x, ya + yb), dim3(256), wlds, st, pj);
            else if (a->wg.action || (b && b->wg.action)) hipLaunchKernelGGL((k_lc_wgrad<true, false>), dim3(x, ya + yb), dim3(256), wlds, st, pj);
            else hipLaunchKernelGGL((k_lc_wgrad<false, false>), dim3(x, ya + yb), dim3(256), wlds, st, pj);
            break;
        }
        case OP_WGRAD_ACT: {
            if (b) { h->bad_dispatch = true; return MZL_E_INVALID; }
            hipLaunchKernelGGL(k_lc_wgrad_act, dim3(a->wa.cout, a->wa.nchunk), dim3(256), 0, st, a->wa);
            hipLaunchKernelGGL(k_lc_wgrad_act_reduce, dim3(cdiv(a->wa.cout * a->wa.A * 9, 256)), dim3(256), 0, st, a->wa);
            break;
        }
        case OP_WREDUCE: {
            Pair<LcWreduce> pj{};
            pj.a = a->wr;
            const int ya = cdiv(a->wr.cout * a->wr.cin, 64), yb = b ? cdiv(b->wr.cout * b->wr.cin, 64) : 0;
            pj.na = ya;
            if (b) pj.b = b->wr;
            hipLaunchKernelGGL(k_lc_wreduce, dim3(1, ya + yb), dim3(64), 0, st, pj);
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


// ---------------------------------------------------------------------------------------------------------------------------------
// MuZeroAtariNet's representation net (network.py:312-353) on the tile path (mz_learn_conv.h, LcTileGather): x -> conv_1 (stride 2) -> relu ->
// res_blocks_1 @ 48 x 48 x 128 -> conv_2 (stride 2) -> relu -> res_blocks_2 @ 24 x 24 -> avg pool -> res_blocks_3 @ 12 x 12 (whole images: the
// fused path of the towers) -> avg pool -> 6 x 6.  Launched one job at a time (nothing to pair with); the planes are cut into 12 x 12 tiles.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int TILE = 12;

struct AtariRun {
    mzlc_learner* h;
    int B;
    hipStream_t st;

    void run(const Op& o) const { launch_ops(h, &o, nullptr, st); }
    // tiles are 12 rows x 12 or 16 columns of plane positions: 16 where the plane's width allows (48: three columns of tiles; 24 does not divide)
    int tile_w(int W) const { return (h->wide_tiles && W % 16 == 0 && W / 16 >= 3) ? 16 : TILE; }
    const Geom& tgeom(int W) const { return tile_w(W) == 16 ? h->gt16 : h->gt; }
    int ntiles(int H, int W) const { return (H / TILE) * (W / tile_w(W)); }
    // plane (H x W, a parity plane when sy == 2) of src [B][C][srcH][srcW] -> tiles with halo
    void gather(const float* src0, const float* src1, const float* coef, int mode, int C, int H, int W, int srcH, int srcW, int sy, int sx, int py, int px,
                int inner_only, float* dst) const {
        LcTileGather g{};
        const int tw = tile_w(W), ts2 = (TILE + 2) * (tw + 2);
        g.src0 = src0; g.src1 = src1; g.coef = coef; g.dst = dst; g.mode = mode; g.B = B; g.C = C; g.cpad = pad16(C); g.H = H; g.W = W;
        g.srcH = srcH; g.srcW = srcW; g.sy = sy; g.sx = sx; g.py = py; g.px = px; g.Ty = TILE; g.Tx = tw; g.nty = H / TILE; g.ntx = W / tw; g.inner_only = inner_only;
        g.n = (long long)B * g.nty * g.ntx * C * ts2;
        static const bool xcd = !getenv("MZLC_NO_GATHER_XCD");
        g.xcd = xcd ? 1 : 0;
        const dim3 grid((unsigned)(B * g.nty * g.ntx), (unsigned)cdiv(C * ts2, 1024));
        if (tw == 16) hipLaunchKernelGGL((k_lc_tile_gather<TILE + 2, 18>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((k_lc_tile_gather<TILE + 2, TILE + 2>), grid, dim3(256), 0, st, g);
    }
    // returns the number of statistic groups written (0 without stat_part)
    int scatter(const float* src, int C, int H, int W, float* dst, int dstH, int dstW, int sy, int sx, int py, int px, const float* skip, float* stat_part,
                bool inner_src = false) const {
        LcTileScatter g{};
        g.src_inner = inner_src ? 1 : 0;
        const int tw = tile_w(W);
        g.src = src; g.dst = dst; g.skip = skip; g.stat_part = stat_part; g.B = B; g.C = C; g.cpad = pad16(C); g.H = H; g.W = W; g.dstH = dstH; g.dstW = dstW;
        g.sy = sy; g.sx = sx; g.py = py; g.px = px; g.Ty = TILE; g.Tx = tw; g.nty = H / TILE; g.ntx = W / tw;
        const dim3 grid(cdiv(H * W, 32), B);
        const int cpt = cdiv(C, 8);
        if (cpt <= 2) hipLaunchKernelGGL(k_lc_tile_scatter<2>, grid, dim3(256), 0, st, g);
        else if (cpt <= 8) hipLaunchKernelGGL(k_lc_tile_scatter<8>, grid, dim3(256), 0, st, g);
        else if (cpt <= 16) hipLaunchKernelGGL(k_lc_tile_scatter<16>, grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL(k_lc_tile_scatter<32>, grid, dim3(256), 0, st, g);
        return B * cdiv(H * W, 32);
    }
    Sched tiles(int C, int H, int W) const { return Sched{h, B * ntiles(H, W), 0, false, tgeom(W), C}; }
    Sched plain(int C) const { return Sched{h, B, 0, false, h->gt, C}; }  // (finalize kernels: the geometry is not used)
    // stride-1 conv of the tiles in `in` -> `out` (forward or data-gradient copy of layer L), identity staging, no epilogue extras
    // (round 6: halo_in -- the tile WITH its halo is the staged slab, only the inner 12 x 12 / 12 x 16 outputs are computed: 9 / 12 pixel tiles
    // per workgroup instead of 13 / 16, written as inner-only tiles [tiles][C][Ty Tx]; MZLC_NO_HALO_IN=1 at create: the first form)
    bool halo_in() const { return h->halo_in; }
    void conv_tiles(const LayerInfo& L, bool dgrad, int H, int W, const float* in, float* out) const {
        const Sched s = tiles(dgrad ? L.cout : L.cin_real, H, W);
        LcConv c = s.conv_base(L, dgrad);
        c.in0 = in; c.in_mode = IN_IDENT; c.out = out; c.stat_mode = ST_NONE;
        Op o = s.op_conv(c);
        if (halo_in()) {
            const int tw = tile_w(W);
            o.conv.halo_in = 1; o.conv.G = 1; o.conv.h = TILE; o.conv.w_img = tw; o.conv.qstride = (4 * (TILE + 2) * (tw + 2) + 63) & ~63;
            o.npt = (TILE * tw) / 16;  // 9 (12 x 12) or 12 (12 x 16)
            o.side15 = 0;
        }
        run(o);
    }
    // (round 6: out_plane) the same conv with the scatter folded into its epilogue: the tile's inner outputs go straight to their place in `plane`
    // [B][C][H][W] (+ `skip`, a plane of the same shape), and with `stat_part` the BatchNorm forward partial sums are taken per tile (pivoted, the
    // format k_lc_bn_fwd reads).  Returns the statistic groups written, or -1 when this build / shape does not take the route (the caller scatters).
    bool plane_route(int C, int H, int W) const { return h->out_plane && (size_t)B * C * H * W * sizeof(float) < ((size_t)1 << 32); }
    // `mask` (with `partner`): the data gradient's consumer folded in as in the towers -- the result is zeroed where the plane `mask` (a materialised
    // ReLU output) is <= 0 and the BatchNorm-backward partial sums against `partner` (that layer's raw conv output) are taken per tile: dz of the
    // layer below leaves the conv directly (before: a gradient plane, then k_lc_entry_plain over it).
    int conv_tiles_plane(const LayerInfo& L, bool dgrad, int H, int W, const float* in, float* plane, const float* skip, float* stat_part,
                         const float* mask = nullptr, const float* partner = nullptr, const float* mcoef = nullptr) const {
        const Sched s = tiles(dgrad ? L.cout : L.cin_real, H, W);
        LcConv c = s.conv_base(L, dgrad);
        c.in0 = in; c.in_mode = IN_IDENT; c.out = plane; c.skip = skip;
        c.stat_mode = stat_part ? (mask ? ST_BWD : ST_FWD) : ST_NONE; c.stat_part = stat_part;
        c.mask = mask; c.mcoef = mcoef; c.partner = partner;  // (mcoef: the mask is where a mask + b > 0 -- a ReLU that was never materialised)
        Op o = s.op_conv(c);
        const int tw = tile_w(W);
        o.conv.halo_in = 1; o.conv.G = 1; o.conv.h = TILE; o.conv.w_img = tw; o.conv.qstride = (4 * (TILE + 2) * (tw + 2) + 63) & ~63;
        o.conv.out_plane = 1; o.conv.pl_h = H; o.conv.pl_w = W; o.conv.pl_nty = H / TILE; o.conv.pl_ntx = W / tw;
        o.npt = (TILE * tw) / 16;
        o.side15 = 0;
        run(o);
        return B * ntiles(H, W);
    }
    // one parity plane's share of a stride-2 conv: the packed copy at `w_off`; accumulate: out += (the earlier planes' sum rides in `skip`)
    void conv_par(int w_off, int cin, int cout, int H, int W, const float* in, float* out, bool accumulate, int tapmask) const {
        const Sched s = tiles(cin, H, W);
        const Geom& tg = tgeom(W);
        LcConv c{};
        c.B = B * ntiles(H, W); c.G = tg.G; c.h = tg.h; c.w_img = tg.w; c.qstride = tg.qstride;
        c.cin_real = cin; c.cin = cin; c.n_cb = cdiv(cin, 16); c.cout = cout; c.co_tiles = cdiv(cout, 16); c.w = h->packed + w_off;
        c.cpad_in = pad16(cin); c.cpad_out = pad16(cout); c.num_actions = h->A;
        c.in0 = in; c.in_mode = IN_IDENT; c.out = out; c.skip = accumulate ? out : nullptr; c.stat_mode = ST_NONE;
        c.tapmask = h->par_compact ? tapmask : 0;
        run(s.op_conv(c));
    }
    void bn_fwd(const LayerInfo& L, float* fcoef, float* save, int groups, float count) const {
        const Sched s = plain(L.cout);
        Op o = s.op_bnfwd(L, fcoef, save);
        o.bf.groups = groups; o.bf.count = count;
        run(o);
    }
    void bn_bwd(const LayerInfo& L, const float* save, float* bcoef, int groups, float count) const {
        const Sched s = plain(L.cout);
        Op o = s.op_bnbwd(L, save, bcoef, groups, 0);
        o.bb.count = count;
        run(o);
    }
    void apply(const float* y, const float* res, const float* coef, float* out, int C, int hw, int cpad = 0) const {
        Op o{};
        o.kind = OP_APPLY;
        o.ap.y = y; o.ap.res = res; o.ap.coef = coef; o.ap.out = out; o.ap.C = C; o.ap.hw = hw; o.ap.cpad = cpad ? cpad : pad16(C); o.ap.n = (long long)B * C * hw;
        run(o);
    }
    // dz = extra [x > 0] + the BatchNorm-backward partial sums against `partner`; returns the groups written to stat[0]
    int entry(const float* x, const float* gs, float scale, const float* extra, const float* partner, float* dz, int C, int hw) const {
        LcEntry e{};
        e.x = x; e.gs = gs; e.extra = extra; e.partner = partner; e.dz = dz; e.stat_part = h->stat[0]; e.scale = scale; e.B = B; e.C = C; e.hw = hw; e.cpad = pad16(C);
        const int cpt = cdiv(C, 8);
        if (!gs && extra && hw % 4 == 0) {  // the tiled stages: four positions per lane
            const int nchunks = cdiv(hw / 4, 32), split = nchunks < 32 ? nchunks : 32;
            const dim3 grid(split, B);
            if (cpt <= 2) hipLaunchKernelGGL(k_lc_entry_plain<2>, grid, dim3(256), 0, st, e);
            else if (cpt <= 8) hipLaunchKernelGGL(k_lc_entry_plain<8>, grid, dim3(256), 0, st, e);
            else if (cpt <= 16) hipLaunchKernelGGL(k_lc_entry_plain<16>, grid, dim3(256), 0, st, e);
            else hipLaunchKernelGGL(k_lc_entry_plain<32>, grid, dim3(256), 0, st, e);
            return B * split;
        }
        const int nchunks = cdiv(hw, 32), split = nchunks < 64 ? nchunks : 64;
        const dim3 grid(split, B);
        if (cpt <= 2) hipLaunchKernelGGL(k_lc_entry<2>, grid, dim3(256), 0, st, e);
        else if (cpt <= 8) hipLaunchKernelGGL(k_lc_entry<8>, grid, dim3(256), 0, st, e);
        else if (cpt <= 16) hipLaunchKernelGGL(k_lc_entry<16>, grid, dim3(256), 0, st, e);
        else hipLaunchKernelGGL(k_lc_entry<32>, grid, dim3(256), 0, st, e);
        return B * split;
    }
    // weight gradient of layer L from dy tiles (already BatchNorm-backward transformed, halo included: ring_zero) and x tiles
    void wgrad_tiles(const LayerInfo& L, int cin, int H, int W, const float* dy_tiles, const float* x_tiles, const signed char* tapmap) const {
        const Sched s = tiles(L.cout, H, W);
        LayerInfo Lw = L;
        Lw.cin_real = cin; Lw.cin = cin;
        std::vector<Op> ops;
        s.wgrad_ops(ops, Lw, dy_tiles, dy_tiles, h->coef_ident, x_tiles, IN_IDENT, nullptr, nullptr, 0);
        ops[0].wg.ring_zero = 1;
        ops[0].wg.cpad_out = pad16(h->P > 128 ? h->P : 128);  // (the stride of coef_ident's rows)
        if (h->ring_rows && ops[0].wg.sg == 1) {  // the dy plane without the tile's first and last row (all ring): h - 2 rows of positions to reduce over
            LcWgrad& g = ops[0].wg;
            g.ring_rows = 1;
            g.nsteps = cdiv((g.h - 2) * g.P4, 16);
            g.SPY = 16 * g.nsteps + 4;
            g.SPX = 2 * g.P4 + 16 * g.nsteps + 12;
            if (h->row_steps && g.P4 > 16 && g.w_img - 2 <= 16 && (g.h - 2) * g.P4 + 4 <= g.SPY) {  // one step per tile row (the planes keep their size)
                g.ring_rows = 2;
                g.nsteps = g.h - 2;
            } else if (h->quad_steps && g.P4 == 16 && g.w_img - 2 == 12 && (3 * (g.h - 2)) % 4 == 0 && (g.h - 2) * g.P4 + 4 <= g.SPY) {  // four real quads per step
                g.ring_rows = 3;
                g.nsteps = 3 * (g.h - 2) / 4;
            }
        }
        if (tapmap) {
            ops[1].wr.use_map = 1;
            int mask = 0;
            for (int t = 0; t < 9; t++) {
                ops[1].wr.tapmap[t] = tapmap[t];
                if (tapmap[t] >= 0) mask |= 1 << t;
            }
            if (h->par_compact) ops[0].wg.tapmask = mask;
        }
        run(ops[0]);
        run(ops[1]);
    }

    // ---- forward of two residual blocks on a tiled stage ----
    float* stage_fwd(mzlc_learner::StageBufs& sb, const std::vector<int>& lay, const float* x_in, int C, int H, int W) const {
        const int hw = H * W;
        const float count = (float)B * (float)hw;
        const float* cur = x_in;
        for (int r = 0; r < 2; r++) {
            const LayerInfo &L1 = h->layers[lay[2 * r]], &L2 = h->layers[lay[2 * r + 1]];
            float* xa = h->keep_tiles ? sb.xt[2 * r] : h->TA;      // (kept: the backward pass's weight gradient multiplies these very tiles)
            float* xb = h->keep_tiles ? sb.xt[2 * r + 1] : h->TA;
            gather(cur, nullptr, nullptr, IN_IDENT, C, H, W, H, W, 1, 1, 0, 0, 0, xa);
            int ng;
            if (plane_route(C, H, W)) ng = conv_tiles_plane(L1, false, H, W, xa, sb.y[2 * r], nullptr, h->stat[0]);
            else {
                conv_tiles(L1, false, H, W, xa, h->TB);
                ng = scatter(h->TB, C, H, W, sb.y[2 * r], H, W, 1, 1, 0, 0, nullptr, h->stat[0], halo_in());
            }
            bn_fwd(L1, sb.fcoef[2 * r], sb.save[2 * r], ng, count);
            if (h->skip_h1 && plane_route(C, H, W)) gather(sb.y[2 * r], nullptr, sb.fcoef[2 * r], IN_BNRELU, C, H, W, H, W, 1, 1, 0, 0, 0, xb);
            else {
                apply(sb.y[2 * r], nullptr, sb.fcoef[2 * r], sb.h1[r], C, hw);
                gather(sb.h1[r], nullptr, nullptr, IN_IDENT, C, H, W, H, W, 1, 1, 0, 0, 0, xb);
            }
            if (plane_route(C, H, W)) ng = conv_tiles_plane(L2, false, H, W, xb, sb.y[2 * r + 1], nullptr, h->stat[0]);
            else {
                conv_tiles(L2, false, H, W, xb, h->TB);
                ng = scatter(h->TB, C, H, W, sb.y[2 * r + 1], H, W, 1, 1, 0, 0, nullptr, h->stat[0], halo_in());
            }
            bn_fwd(L2, sb.fcoef[2 * r + 1], sb.save[2 * r + 1], ng, count);
            apply(sb.y[2 * r + 1], cur, sb.fcoef[2 * r + 1], sb.x[r], C, hw);
            cur = sb.x[r];
        }
        return const_cast<float*>(cur);
    }
    // ---- backward: sb.dzA holds dz of the stage's last BatchNorm, its partial sums (ng groups) are in stat[0]; leaves the gradient wrt x_in in sb.gF ----
    void stage_bwd(mzlc_learner::StageBufs& sb, const std::vector<int>& lay, const float* x_in, int C, int H, int W, int ng) const {
        const int hw = H * W;
        const float count = (float)B * (float)hw;
        for (int r = 1; r >= 0; r--) {
            const LayerInfo &L1 = h->layers[lay[2 * r]], &L2 = h->layers[lay[2 * r + 1]];
            const float* xin_blk = r > 0 ? sb.x[r - 1] : x_in;
            bn_bwd(L2, sb.save[2 * r + 1], sb.bcoef[2 * r + 1], ng, count);
            gather(sb.dzA, sb.y[2 * r + 1], sb.bcoef[2 * r + 1], IN_BNBWD, C, H, W, H, W, 1, 1, 0, 0, 0, h->TA);
            if (!h->keep_tiles) gather(sb.h1[r], nullptr, nullptr, IN_IDENT, C, H, W, H, W, 1, 1, 0, 0, 0, h->TC);
            wgrad_tiles(L2, C, H, W, h->TA, h->keep_tiles ? sb.xt[2 * r + 1] : h->TC, nullptr);
            if (plane_route(C, H, W) && h->skip_h1) ng = conv_tiles_plane(L2, true, H, W, h->TA, sb.dzB, nullptr, h->stat[0], sb.y[2 * r], sb.y[2 * r], sb.fcoef[2 * r]);
            else if (plane_route(C, H, W) && h->fuse_entry) ng = conv_tiles_plane(L2, true, H, W, h->TA, sb.dzB, nullptr, h->stat[0], sb.h1[r], sb.y[2 * r]);
            else {
                if (plane_route(C, H, W)) conv_tiles_plane(L2, true, H, W, h->TA, sb.gF, nullptr, nullptr);
                else {
                    conv_tiles(L2, true, H, W, h->TA, h->TB);
                    scatter(h->TB, C, H, W, sb.gF, H, W, 1, 1, 0, 0, nullptr, nullptr, halo_in());
                }
                ng = entry(sb.h1[r], nullptr, 1.0f, sb.gF, sb.y[2 * r], sb.dzB, C, hw);
            }
            bn_bwd(L1, sb.save[2 * r], sb.bcoef[2 * r], ng, count);
            gather(sb.dzB, sb.y[2 * r], sb.bcoef[2 * r], IN_BNBWD, C, H, W, H, W, 1, 1, 0, 0, 0, h->TA);
            if (!h->keep_tiles) gather(xin_blk, nullptr, nullptr, IN_IDENT, C, H, W, H, W, 1, 1, 0, 0, 0, h->TC);
            wgrad_tiles(L1, C, H, W, h->TA, h->keep_tiles ? sb.xt[2 * r] : h->TC, nullptr);
            if (plane_route(C, H, W) && h->fuse_entry && r > 0)  // (in place: every element of dzA is read -- the skip gradient -- and written by the same lane)
                ng = conv_tiles_plane(L1, true, H, W, h->TA, sb.dzA, sb.dzA, h->stat[0], xin_blk, sb.y[2 * r - 1]);
            else {
                if (plane_route(C, H, W)) conv_tiles_plane(L1, true, H, W, h->TA, sb.gF, sb.dzA, nullptr);  // + the block's skip gradient
                else {
                    conv_tiles(L1, true, H, W, h->TA, h->TB);
                    scatter(h->TB, C, H, W, sb.gF, H, W, 1, 1, 0, 0, sb.dzA, nullptr, halo_in());  // + the block's skip gradient
                }
                if (r > 0) ng = entry(xin_blk, nullptr, 1.0f, sb.gF, sb.y[2 * r - 1], sb.dzA, C, hw);
            }
        }
    }
};

// tap maps of the parity planes (mz_learn_conv.h, LcTileGather).  Forward / weight gradient: stride-1 tap (sy, sx) of plane (p, q) is weight tap
// (krow(p, sy), krow(q, sx)); data gradient of a stride-2 conv: output parity plane (p, q) of dx from dy.
int par_row(int p, int s) { return p == 0 ? (s == 1 ? 1 : -1) : (s == 0 ? 0 : (s == 1 ? 2 : -1)); }
int par_row_d(int p, int s) { return p == 0 ? (s == 1 ? 1 : -1) : (s == 2 ? 0 : (s == 1 ? 2 : -1)); }
void par_tapmap(int p, int q, bool dgrad, signed char* m) {
    for (int sy = 0; sy < 3; sy++)
        for (int sx = 0; sx < 3; sx++) {
            const int ky = dgrad ? par_row_d(p, sy) : par_row(p, sy), kx = dgrad ? par_row_d(q, sx) : par_row(q, sx);
            m[sy * 3 + sx] = (signed char)((ky >= 0 && kx >= 0) ? ky * 3 + kx : -1);
        }
}

int par_tapmask(int p, int q, bool dgrad) {
    signed char m[9];
    par_tapmap(p, q, dgrad, m);
    int mask = 0;
    for (int t = 0; t < 9; t++)
        if (m[t] >= 0) mask |= 1 << t;
    return mask;
}

// forward of the Atari representation; returns the 6 x 6 raw hidden state
float* atari_rep_fwd(mzlc_learner* h, int B, hipStream_t st) {
    const AtariRun R{h, B, st};
    const int H1 = h->obsH / 2, W1 = h->obsW / 2, H2 = H1 / 2, W2 = W1 / 2, H3 = H2 / 2, W3 = W2 / 2;
    const LayerInfo &C1 = h->layers[h->l_c1], &C2 = h->layers[h->l_c2];
    // conv_1: four parity planes of the observation, accumulated on the tiles
    for (int pq = 0; pq < 4; pq++) {
        R.gather(h->obs, nullptr, nullptr, IN_IDENT, h->C0, H1, W1, h->obsH, h->obsW, 2, 2, pq >> 1, pq & 1, 0, h->TA);
        R.conv_par(h->par_f[0][pq], h->C0, C1.cout, H1, W1, h->TA, h->TB, pq > 0, par_tapmask(pq >> 1, pq & 1, false));
    }
    R.scatter(h->TB, C1.cout, H1, W1, h->y_c1, H1, W1, 1, 1, 0, 0, nullptr, nullptr);
    const int big = pad16(h->P > 128 ? h->P : 128);  // the row stride of coef_relu / coef_ident
    R.apply(h->y_c1, nullptr, h->coef_relu, h->a1, C1.cout, H1 * W1, big);
    float* x48 = R.stage_fwd(h->sb48, h->l_b1, h->a1, 128, H1, W1);
    for (int pq = 0; pq < 4; pq++) {
        R.gather(x48, nullptr, nullptr, IN_IDENT, 128, H2, W2, H1, W1, 2, 2, pq >> 1, pq & 1, 0, h->TA);
        R.conv_par(h->par_f[1][pq], 128, C2.cout, H2, W2, h->TA, h->TB, pq > 0, par_tapmask(pq >> 1, pq & 1, false));
    }
    R.scatter(h->TB, h->P, H2, W2, h->y_c2, H2, W2, 1, 1, 0, 0, nullptr, nullptr);
    R.apply(h->y_c2, nullptr, h->coef_relu, h->a2, h->P, H2 * W2, big);
    float* x24 = R.stage_fwd(h->sb24, h->l_b2, h->a2, h->P, H2, W2);
    {
        const long long n = (long long)B * h->P * H3 * W3;
        hipLaunchKernelGGL(k_lc_pool_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x24, h->p1, n, H2, W2, H3, W3);
    }
    std::vector<Op> ops;
    const Sched s12{h, B, 0, false, h->g12, h->P};
    float* x12 = s12.tower_fwd(ops, h->t12, h->a12, h->p1, nullptr);
    for (const Op& o : ops) launch_ops(h, &o, nullptr, st);
    {
        const long long n = (long long)B * h->P * h->hw;
        hipLaunchKernelGGL(k_lc_pool_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x12, h->hraw, n, H3, W3, h->h, h->w);
    }
    return h->hraw;
}

// backward of the Atari representation from the gradient wrt the normalised hidden state s_0
void atari_rep_bwd(mzlc_learner* h, int B, const float* gs0, hipStream_t st) {
    const AtariRun R{h, B, st};
    const int H1 = h->obsH / 2, W1 = h->obsW / 2, H2 = H1 / 2, W2 = W1 / 2, H3 = H2 / 2, W3 = W2 / 2;
    const LayerInfo &C1 = h->layers[h->l_c1], &C2 = h->layers[h->l_c2];
    const int P = h->P;
    float* x12 = h->a12.x.back();
    float* x24 = h->sb24.x[1];
    float* x48 = h->sb48.x[1];
    // normalize backward (the mask x > 0 of k_lc_entry is harmless on an average of ReLU outputs: where it is 0 every input of the window is 0 and masked
    // below), average-pool backward, mask + partial sums of res_blocks_3's last BatchNorm
    R.entry(h->hraw, gs0, 1.0f, nullptr, h->hraw, h->dH, P, h->hw);
    {
        const long long n = (long long)B * P * H3 * W3;
        hipLaunchKernelGGL(k_lc_pool_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h->dH, h->g12in, n, H3, W3, h->h, h->w);
    }
    {
        LcEntry e{};
        float* keep[3] = {h->D[0][0], h->D[0][1], h->D[0][2]};
        for (int i = 0; i < 3; i++) h->D[0][i] = h->D12[i];  // (the op builders read the buffers of lane 0; restored below)
        e.x = x12; e.gs = nullptr; e.extra = h->g12in; e.partner = h->a12.y.back(); e.dz = h->D[0][0]; e.stat_part = h->stat[0]; e.scale = 1.0f;
        e.B = B; e.C = P; e.hw = H3 * W3; e.cpad = pad16(P);
        const int nchunks = cdiv(e.hw, 32), split = nchunks < ENTRY_SPLIT ? nchunks : ENTRY_SPLIT;
        const dim3 grid(split, B);
        const int cpt = cdiv(P, 8);
        if (cpt <= 2) hipLaunchKernelGGL(k_lc_entry<2>, grid, dim3(256), 0, st, e);
        else if (cpt <= 8) hipLaunchKernelGGL(k_lc_entry<8>, grid, dim3(256), 0, st, e);
        else if (cpt <= 16) hipLaunchKernelGGL(k_lc_entry<16>, grid, dim3(256), 0, st, e);
        else hipLaunchKernelGGL(k_lc_entry<32>, grid, dim3(256), 0, st, e);
        std::vector<Op> ops;
        const Sched s12{h, B, 0, false, h->g12, P};
        s12.tower_bwd(ops, h->t12, h->a12, h->p1, nullptr, B * split, 0, h->g12in, nullptr);  // gradient wrt the pooled 12 x 12 input -> g12in
        for (const Op& o : ops) launch_ops(h, &o, nullptr, st);
        for (int i = 0; i < 3; i++) h->D[0][i] = keep[i];
    }
    // 24 x 24 stage
    {
        const long long n = (long long)B * P * H2 * W2;
        hipLaunchKernelGGL(k_lc_pool_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h->g12in, h->sb24.gF, n, H2, W2, H3, W3);
    }
    int ng = R.entry(x24, nullptr, 1.0f, h->sb24.gF, h->sb24.y[3], h->sb24.dzA, P, H2 * W2);
    R.stage_bwd(h->sb24, h->l_b2, h->a2, P, H2, W2, ng);
    // conv_2 (stride 2, relu without BatchNorm): dy = g [a2 > 0]
    {
        const long long n = (long long)B * P * H2 * W2;
        hipLaunchKernelGGL(k_lc_relu_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h->sb24.gF, h->a2, h->sb24.dzB, n);
    }
    R.gather(h->sb24.dzB, nullptr, nullptr, IN_IDENT, P, H2, W2, H2, W2, 1, 1, 0, 0, 0, h->TA);  // dy tiles with halo: both gradients read them
    for (int pq = 0; pq < 4; pq++) {
        signed char m[9];
        par_tapmap(pq >> 1, pq & 1, false, m);
        R.gather(x48, nullptr, nullptr, IN_IDENT, 128, H2, W2, H1, W1, 2, 2, pq >> 1, pq & 1, 0, h->TC);
        R.wgrad_tiles(C2, 128, H2, W2, h->TA, h->TC, m);
    }
    for (int pq = 0; pq < 4; pq++) {  // data gradient: parity plane (p, q) of the 48 x 48 gradient from the dy tiles
        R.conv_par(h->par_d[pq], P, 128, H2, W2, h->TA, h->TB, false, par_tapmask(pq >> 1, pq & 1, true));
        R.scatter(h->TB, 128, H2, W2, h->sb48.gF, H1, W1, 2, 2, pq >> 1, pq & 1, nullptr, nullptr);
    }
    ng = R.entry(x48, nullptr, 1.0f, h->sb48.gF, h->sb48.y[3], h->sb48.dzA, 128, H1 * W1);
    R.stage_bwd(h->sb48, h->l_b1, h->a1, 128, H1, W1, ng);
    // conv_1: weight gradient only (its input is the observation)
    {
        const long long n = (long long)B * 128 * H1 * W1;
        hipLaunchKernelGGL(k_lc_relu_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h->sb48.gF, h->a1, h->sb48.dzB, n);
    }
    R.gather(h->sb48.dzB, nullptr, nullptr, IN_IDENT, 128, H1, W1, H1, W1, 1, 1, 0, 0, 0, h->TA);
    for (int pq = 0; pq < 4; pq++) {
        signed char m[9];
        par_tapmap(pq >> 1, pq & 1, false, m);
        R.gather(h->obs, nullptr, nullptr, IN_IDENT, h->C0, H1, W1, h->obsH, h->obsW, 2, 2, pq >> 1, pq & 1, 0, h->TC);
        R.wgrad_tiles(C1, h->C0, H1, W1, h->TA, h->TC, m);
    }
}

}  // namespace

// =================================================================================================================================
int mzlc_create(const mzl_config* cfg, int device_id, int num_cus, mzlc_learner** out, std::string& err) {
    mzlc_learner* h = new mzlc_learner();
    h->cfg = *cfg; h->device = device_id; h->num_cus = num_cus > 0 ? num_cus : 256;
    h->P = cfg->num_planes; h->C0 = cfg->in_channels; h->A = cfg->num_actions; h->R = cfg->num_res_blocks; h->K = cfg->unroll_steps;
    h->atari = cfg->net_kind == MZL_NET_ATARI;
    h->maxB = cfg->max_batch;
    h->paired = !getenv("MZLC_NO_PAIR");
    h->fuse_apply = !getenv("MZLC_NO_FUSE_APPLY");
    h->xcd_remap = !getenv("MZLC_NO_XCD_REMAP");
    h->par_compact = !getenv("MZLC_NO_TAPSETS");
    h->halo_in = !getenv("MZLC_NO_HALO_IN");
    h->ring_rows = !getenv("MZLC_NO_RING_ROWS");
    h->keep_tiles = !getenv("MZLC_NO_KEEP_TILES");
    h->out_plane = h->halo_in && !getenv("MZLC_NO_OUT_PLANE");
    h->row_steps = h->ring_rows && !getenv("MZLC_NO_ROW_STEPS");
    h->quad_steps = h->ring_rows && !getenv("MZLC_NO_QUAD_STEPS");
    h->fuse_entry = !getenv("MZLC_NO_FUSE_ENTRY");
    h->skip_h1 = h->out_plane && h->fuse_entry && h->keep_tiles && !getenv("MZLC_KEEP_H1");
    h->act_sparse = !getenv("MZLC_ACT_MFMA") && h->A <= 256;
    if (const char* m = getenv("MZLC_WGRAD_MIN_IPW")) h->wgrad_min_ipw = atoi(m) > 0 ? atoi(m) : 1;
    auto bad = [&](const std::string& m) { err = m; mzlc_destroy(h); return MZL_E_INVALID; };
    if (h->atari) {  // the observation is board_h x board_w (96 x 96 in every reference configuration); the hidden state 1 / 16 of it
        h->obsH = cfg->board_h; h->obsW = cfg->board_w;
        if (h->obsH < 1 || h->obsH % (16 * TILE / 2) || h->obsW % (16 * TILE / 2) || h->obsH != h->obsW)
            return bad("Atari net: square frames whose side is a multiple of 96 (the 48 x 48 and 24 x 24 stages are cut into 12 x 12 tiles)");
        h->h = h->obsH / 16; h->w = h->obsW / 16;
        if (h->obsH / 8 != TILE) return bad("Atari net: 96 x 96 frames (the 12 x 12 stage runs as whole images)");
    } else {
        h->h = cfg->board_h; h->w = cfg->board_w; h->obsH = h->h; h->obsW = h->w;
    }
    h->hw = h->h * h->w;
    h->max_imgs = h->maxB * (h->atari ? (h->obsH / 2 / TILE) * (h->obsW / 2 / TILE) : 1);
    if (h->C0 < 1 || h->h < 1 || h->w < 1 || h->R < 1 || h->P < 1) return bad("bad conv-net geometry (in_channels, board_h, board_w, num_res_blocks, num_planes)");
    if (cfg->in_dim != h->C0 * h->obsH * h->obsW) return bad("in_dim must equal in_channels * board_h * board_w");
    if (cfg->value_support_size < 1 || cfg->reward_support_size < 1 || cfg->value_support_size > 1024 || cfg->reward_support_size > 1024) return bad("support sizes must be in [1, 1024]");
    if (h->hw > 240 || h->P > 1024) return bad("conv learner: boards up to 240 points and 1024 planes (larger nets train through muzero_amd.learner.train_step)");
    if (!make_geom(h->gm, h->h, h->w, !getenv("MZLC_NO_SIDE"), getenv("MZLC_DENSE_TILING") ? 0 : h->maxB, cdiv(cdiv(h->P, 16), 4) * 2, h->num_cus, !getenv("MZLC_NO_WGRAD_STACK"), false,
                   cdiv(cdiv(h->P, 16), 2) * cdiv(cdiv(h->P, 16), 2))) return bad("board does not fit the conv kernels' tiling");
    if (wgrad_lds(h->gm) > 160 * 1024 || conv_lds(h->gm.qstride, pad16(h->P + h->A)) > 160 * 1024) return bad("board too large for the conv learner's LDS layout");
    h->wide_tiles = !getenv("MZLC_NO_WIDE_TILES");
    if (h->atari && (!make_geom(h->gt, TILE + 2, TILE + 2, false) || !make_geom(h->g12, TILE, TILE, false) ||
                     !make_geom(h->gt16, TILE + 2, 18, false, 0, 1, 256, true, true)))
        return bad("internal: tile geometry");
    if (h->atari && (wgrad_lds(h->gt16) > 80 * 1024 || h->gt16.npt != 16)) h->wide_tiles = false;  // (two weight-gradient workgroups per CU need their LDS)
    // ---- parameter / buffer tables in state_dict order (network.py:312-498) ----
    const int kv = cfg->value_support_size > 1 ? 2 : 0, kr = cfg->reward_support_size > 1 ? 2 : 0;  // head kinds: categorical (2-hot cross entropy) | squared error
    if (h->atari) {
        auto add_plain = [&](const std::string& name, int cin, int cout) {  // stride-2 conv without BatchNorm: parity copies instead of the standard ones
            LayerInfo L;
            L.cin_real = cin; L.cin = cin; L.cout = cout; L.cin_d = 0; L.co_tiles = cdiv(cout, 16); L.n_cb = cdiv(cin, 16); L.f_off = -1;
            add_tensor(h, name + ".weight", cout, cin * 9, &L.w_off);
            h->layers.push_back(L);
            return (int)h->layers.size() - 1;
        };
        auto add_blocks = [&](const std::string& pre, int C, std::vector<int>& out) {
            for (int r = 0; r < 2; r++) {
                const std::string b = pre + "." + std::to_string(r);
                out.push_back(add_conv(h, b + ".conv_block1.0", b + ".conv_block1.1", C, 0, C, C));
                out.push_back(add_conv(h, b + ".conv_block2.0", b + ".conv_block2.1", C, 0, C, C));
            }
        };
        h->l_c1 = add_plain("represent_net.conv_1", h->C0, 128);
        add_blocks("represent_net.res_blocks_1", 128, h->l_b1);
        h->l_c2 = add_plain("represent_net.conv_2", 128, h->P);
        add_blocks("represent_net.res_blocks_2", h->P, h->l_b2);
        add_blocks("represent_net.res_blocks_3", h->P, h->t12.layers);
        h->t12.conv0 = false; h->t12.R = 2;
        for (int k = 0; k < 2; k++) {  // parity copies: forward of conv_1 / conv_2, data gradient of conv_2
            const LayerInfo& L = h->layers[k == 0 ? h->l_c1 : h->l_c2];
            for (int pq = 0; pq < 4; pq++) { h->par_f[k][pq] = (int)h->packed_floats; h->packed_floats += (size_t)L.co_tiles * L.n_cb * 9 * 256; }
        }
        for (int pq = 0; pq < 4; pq++) { h->par_d[pq] = (int)h->packed_floats; h->packed_floats += (size_t)cdiv(128, 16) * cdiv(h->P, 16) * 9 * 256; }
    } else {
        add_tower(h, 0, "represent_net", true, h->C0, 0, false);
    }
    add_tower(h, 1, "dynamics_net", true, h->P, h->A, true);
    add_head(h, 0, "dynamics_net.reward_head", 1, cfg->reward_support_size, kr);
    add_tower(h, 2, "prediction_net", false, 0, 0, true);
    add_head(h, 1, "prediction_net.policy_net", 2, h->A, 1);
    add_head(h, 2, "prediction_net.value_net", 1, cfg->value_support_size, kv);
    h->n_max = h->A;
    if (cfg->value_support_size > h->n_max) h->n_max = cfg->value_support_size;
    if (cfg->reward_support_size > h->n_max) h->n_max = cfg->reward_support_size;
    // ---- device memory ----
    h->T = (size_t)h->maxB * h->P * h->hw;
    bool ok = true;
    auto AL = [&](float** p, size_t n) { ok = ok && dalloc(h, p, n) == hipSuccess; };
    AL(&h->packed, h->packed_floats);
    {
        std::vector<LcPackJob> jobs;
        for (const LayerInfo& L : h->layers) {
            if (L.f_off < 0) continue;  // (the Atari net's stride-2 convs: parity copies, below)
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
    if (!h->atari) ok = ok && alloc_app(h, h->tower[0], h->app_rep, h->T);
    h->app_dyn.resize(h->K); h->app_pred.resize(h->K);
    {   // one launch per shared layer for the K steps' weight gradients where a single step's share is a few staging rounds per workgroup
        const int blocks = cdiv(cdiv(h->P, 16), 2) * cdiv(cdiv(h->P, 16), 2);
        int chunks = (h->paired ? 1 : 2) * h->num_cus / blocks;
        chunks = chunks < 1 ? 1 : (chunks > h->maxB ? h->maxB : chunks);
        const int rounds = cdiv(cdiv(h->maxB, chunks), h->gm.SG);
        h->defer_wgrad = h->K > 1 && rounds <= 4 && h->tower[1].R > 0;
        if (const char* m = getenv("MZLC_DEFER_WGRAD")) h->defer_wgrad = atoi(m) != 0 && h->K > 1 && h->tower[1].R > 0;
    }
    for (int t = 0; t < h->K; t++)
        ok = ok && alloc_app(h, h->tower[1], h->app_dyn[t], h->T, h->defer_wgrad) && alloc_app(h, h->tower[2], h->app_pred[t], h->T, h->defer_wgrad);
    h->s.resize(h->K);
    for (int t = 0; t < h->K; t++) AL(&h->s[t], h->T);
    if (h->defer_wgrad && ok) {  // the steps' operand table of every block layer (fixed buffers: written once)
        std::vector<LcWgradSrc> tab;
        for (int tw = 0; tw < 2; tw++) {
            const TowerInfo& T = h->tower[1 + tw];
            const int nl = (int)T.layers.size(), li = T.conv0 ? 1 : 0, xi = li;
            h->srcs_stride[tw] = nl;
            for (int l = 0; l < nl; l++)
                for (int t = 0; t < h->K; t++) {
                    const AppBufs& a = tw == 0 ? h->app_dyn[t] : h->app_pred[t];
                    LcWgradSrc e{};
                    if (l >= li) {
                        const int r = (l - li) >> 1;
                        e.dz = a.dz[l]; e.y = a.y[l]; e.dcoef = a.bcoef[l];
                        if ((l - li) & 1) { e.x0 = a.y[l - 1]; e.xcoef = a.fcoef[l - 1]; }  // second conv of block r: x = relu(bn(y1)), formed while staging
                        else { e.x0 = r > 0 ? a.x[xi + r - 1] : (T.conv0 ? a.x[0] : h->s[t]); e.xcoef = nullptr; }
                    }
                    tab.push_back(e);
                }
        }
        ok = ok && dalloc(h, &h->d_srcs, tab.size()) == hipSuccess &&
             hipMemcpy(h->d_srcs, tab.data(), tab.size() * sizeof(LcWgradSrc), hipMemcpyHostToDevice) == hipSuccess;
    }
    AL(&h->obs, (size_t)h->maxB * h->C0 * h->obsH * h->obsW);
    ok = ok && dalloc(h, &h->act, (size_t)h->K * h->maxB) == hipSuccess;
    {
        const int g_conv = cdiv(h->maxB, h->gm.G), g_entry = h->maxB * cdiv(h->hw, 32);
        h->stat_groups_cap = g_conv > g_entry ? g_conv : g_entry;
        if (h->atari) {  // tile scatter / entry kernels of the 48 x 48 stage: one group per 32 positions, and the 12 x 12 stage's conv groups
            const int g_big = h->maxB * cdiv((h->obsH / 2) * (h->obsW / 2), 32);
            h->stat_groups_cap = g_big > h->stat_groups_cap ? g_big : h->stat_groups_cap;
        }
        const int cpad = pad16(h->P > 128 ? h->P : 128);
        for (int l = 0; l < 2; l++) AL(&h->stat[l], (size_t)h->stat_groups_cap * cpad * 4);  // (forward partials: four floats per group and channel)
        // weight-gradient partials: chunks <= min(B, CUs / blocks); chunks * blocks <= max(CUs, blocks)
        size_t mx = 0;
        for (const LayerInfo& L : h->layers) {
            const int cot = L.co_tiles, cit = cdiv(L.cin, 16), blocks = cdiv(cot, 2) * cdiv(cit, 2);
            int chunks = 2 * h->num_cus / blocks;
            chunks = chunks < 1 ? 1 : (chunks > h->max_imgs ? h->max_imgs : chunks);
            if (h->defer_wgrad) {  // (wgrad_ops_steps: at least one chunk per step)
                int cps = 2 * h->num_cus / blocks / h->K;
                cps = cps < 1 ? 1 : (cps > h->maxB ? h->maxB : cps);
                chunks = h->K * cps > chunks ? h->K * cps : chunks;
            }
            const size_t n = (size_t)chunks * 9 * cot * 16 * cit * 16;
            mx = n > mx ? n : mx;
        }
        for (int l = 0; l < 2; l++) AL(&h->wpart[l], mx);
        AL(&h->wpart_act, (size_t)ACT_CHUNKS * h->P * h->A * 9);
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
        AL(&h->hdl, gb * h->n_max); AL(&h->hspart, gb * LCH_MAXOC * 2); AL(&h->hspiv, gb * LCH_MAXOC); AL(&h->hcoef, (size_t)ng * LCH_MAXOC * 5); AL(&h->hsave, (size_t)ng * LCH_MAXOC * 2);
        AL(&h->hlpart, gb);
        for (int i = 0; i < 3; i++) {
            h->hp_off[i] = h->hp_total;
            h->hp_total += h->head[i].oc * h->P + h->head[i].n_out * h->head[i].oc * h->hw + h->head[i].n_out;
        }
        AL(&h->hwpart, (size_t)h->K * h->hp_total);
    }
    if (h->atari) {
        const int H1 = h->obsH / 2, W1 = h->obsW / 2, H2 = H1 / 2, W2 = W1 / 2, H3 = H2 / 2, W3 = W2 / 2, P = h->P;
        const size_t n48 = (size_t)h->maxB * 128 * H1 * W1, n24 = (size_t)h->maxB * P * H2 * W2, n12 = (size_t)h->maxB * P * H3 * W3;
        const int cpad = pad16(P > 128 ? P : 128);
        auto stage = [&](mzlc_learner::StageBufs& sb, size_t n) {
            sb.y.resize(4); sb.fcoef.resize(4); sb.save.resize(4); sb.bcoef.resize(4); sb.h1.resize(2); sb.x.resize(2);
            for (int i = 0; i < 4; i++) { AL(&sb.y[i], n); AL(&sb.fcoef[i], (size_t)3 * cpad); AL(&sb.save[i], (size_t)2 * cpad); AL(&sb.bcoef[i], (size_t)3 * cpad); }
            for (int i = 0; i < 2; i++) { AL(&sb.h1[i], n); AL(&sb.x[i], n); }
            AL(&sb.dzA, n); AL(&sb.dzB, n); AL(&sb.gF, n);
        };
        stage(h->sb48, n48);
        stage(h->sb24, n24);
        if (h->keep_tiles) {  // (288 GB of HBM: the 1.2 GB of tiles per update are re-read once instead of re-gathered)
            const size_t ts2k = (size_t)(TILE + 2) * (TILE + 2);
            const size_t t48 = (size_t)h->maxB * (H1 / TILE) * (W1 / TILE) * 128 * ts2k, t24 = (size_t)h->maxB * (H2 / TILE) * (W2 / TILE) * P * ts2k;
            h->sb48.xt.resize(4); h->sb24.xt.resize(4);
            for (int i = 0; i < 4; i++) { AL(&h->sb48.xt[i], t48); AL(&h->sb24.xt[i], t24); }
        }
        AL(&h->y_c1, n48); AL(&h->a1, n48); AL(&h->y_c2, n24); AL(&h->a2, n24); AL(&h->p1, n12); AL(&h->g12in, n12); AL(&h->hraw, h->T); AL(&h->dH, h->T);
        for (int i = 0; i < 3; i++) AL(&h->D12[i], n12);
        ok = ok && alloc_app(h, h->t12, h->a12, n12);
        const size_t ts2 = (size_t)(TILE + 2) * (TILE + 2);
        size_t nt = (size_t)h->maxB * (H1 / TILE) * (W1 / TILE) * 128 * ts2;
        const size_t nt24 = (size_t)h->maxB * (H2 / TILE) * (W2 / TILE) * P * ts2;
        nt = nt24 > nt ? nt24 : nt;
        AL(&h->TA, nt); AL(&h->TB, nt); AL(&h->TC, nt);
        AL(&h->coef_relu, (size_t)3 * cpad); AL(&h->coef_ident, (size_t)3 * cpad);
        if (ok) {
            std::vector<float> one(3 * cpad, 0.0f);
            for (int c = 0; c < cpad; c++) one[c] = 1.0f;  // row 0 = 1: (a, b) = (1, 0) and (c1, c2, c3) = (1, 0, 0)
            ok = hipMemcpy(h->coef_relu, one.data(), one.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(h->coef_ident, one.data(), one.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
        }
        std::vector<LcPackPar> pj;
        for (int k = 0; k < 2; k++) {
            const LayerInfo& L = h->layers[k == 0 ? h->l_c1 : h->l_c2];
            for (int pq = 0; pq < 4; pq++) {
                LcPackPar j{};
                j.w_off = L.w_off; j.cout = L.cout; j.cin = L.cin; j.dst_off = h->par_f[k][pq]; j.n_cb = L.n_cb; j.co_tiles = L.co_tiles; j.transpose = 0; j.compact = h->par_compact ? 1 : 0;
                par_tapmap(pq >> 1, pq & 1, false, j.tapmap);
                pj.push_back(j);
            }
        }
        {
            const LayerInfo& L = h->layers[h->l_c2];
            for (int pq = 0; pq < 4; pq++) {
                LcPackPar j{};
                j.w_off = L.w_off; j.cout = L.cout; j.cin = L.cin; j.dst_off = h->par_d[pq]; j.n_cb = cdiv(L.cout, 16); j.co_tiles = cdiv(L.cin, 16); j.transpose = 1; j.compact = h->par_compact ? 1 : 0;
                par_tapmap(pq >> 1, pq & 1, true, j.tapmap);
                pj.push_back(j);
            }
        }
        h->n_pack_par = (int)pj.size();
        ok = ok && dalloc(h, &h->d_pack_par, pj.size()) == hipSuccess;
        if (ok) ok = hipMemcpy(h->d_pack_par, pj.data(), pj.size() * sizeof(LcPackPar), hipMemcpyHostToDevice) == hipSuccess;
    }
    h->sq_blocks = (int)((h->total + 1023) / 1024);
    AL(&h->d_sq, (size_t)h->sq_blocks);
    if (!ok) {
        err = "hipMalloc failed (conv learner: " + std::to_string((double)h->T * 4 * (h->tower[0].layers.size() * 1.5 + h->K * 50) / 1e9) + " GB class)";
        mzlc_destroy(h);
        return MZL_E_HIP;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_wgrad<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_wgrad<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_wgrad<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_wgrad<false, true, 0x010>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_wgrad<false, true, 0x018>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_wgrad<false, true, 0x012>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_wgrad<false, true, 0x01b>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = conv_attr<15, 15>();
    if (e == hipSuccess) e = conv_attr<15, 0>();
    if (e == hipSuccess) e = conv_attr<9, 0>();
    if (e == hipSuccess) e = conv_attr<6, 0>();
    if (e == hipSuccess) e = conv_attr<5, 0>();
    if (e == hipSuccess) e = conv_attr<13, 0>();
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_conv<9, IN_IDENT, 0, 0x1ff, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lc_conv<12, IN_IDENT, 0, 0x1ff, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = conv_taps_attr<13, 0x010>();
    if (e == hipSuccess) e = conv_taps_attr<13, 0x018>();
    if (e == hipSuccess) e = conv_taps_attr<13, 0x012>();
    if (e == hipSuccess) e = conv_taps_attr<13, 0x01b>();
    if (e == hipSuccess) e = conv_taps_attr<13, 0x030>();
    if (e == hipSuccess) e = conv_taps_attr<13, 0x090>();
    if (e == hipSuccess) e = conv_taps_attr<13, 0x1b0>();
    if (e == hipSuccess) e = conv_taps_attr<15, 0x010>();
    if (e == hipSuccess) e = conv_taps_attr<15, 0x018>();
    if (e == hipSuccess) e = conv_taps_attr<15, 0x012>();
    if (e == hipSuccess) e = conv_taps_attr<15, 0x01b>();
    if (e == hipSuccess) e = conv_taps_attr<15, 0x030>();
    if (e == hipSuccess) e = conv_taps_attr<15, 0x090>();
    if (e == hipSuccess) e = conv_taps_attr<15, 0x1b0>();
    if (e == hipSuccess) e = conv_taps_attr<16, 0x1ff>();
    if (e == hipSuccess) e = conv_taps_attr<12, 0x1ff>();
    if (e == hipSuccess) e = conv_taps_attr<16, 0x010>();
    if (e == hipSuccess) e = conv_taps_attr<16, 0x018>();
    if (e == hipSuccess) e = conv_taps_attr<16, 0x012>();
    if (e == hipSuccess) e = conv_taps_attr<16, 0x01b>();
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
    if (h->n_pack_par) hipLaunchKernelGGL(k_lc_pack_par, dim3(64, h->n_pack_par), dim3(256), 0, st, h->d_pack_par, h->params, h->packed);
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
    bt.in_dim = h->C0 * h->obsH * h->obsW;
    hipLaunchKernelGGL(k_lc_gather, dim3(cdiv(bt.in_dim, 256) > 8 ? 8 : cdiv(bt.in_dim, 256), B), dim3(256), 0, st, bt, h->obs, h->act);
    Sched sr{h, B, 0, false, h->gm, h->P}, s0{h, B, 0, paired, h->gm, h->P}, s1{h, B, 1, paired, h->gm, h->P};
    // ---- forward ----
    std::vector<Op> ops, ops2;
    float* hraw;
    if (h->atari) {
        hraw = atari_rep_fwd(h, B, st);
    } else {
        hraw = sr.tower_fwd(ops, h->tower[0], h->app_rep, h->obs, nullptr);
        for (const Op& o : ops) launch_ops(h, &o, nullptr, st);
    }
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
    // The table is a function of the handle (fixed activation buffers, heads and unroll steps): it reaches the device when it CHANGES -- the
    // first call, in practice -- not on every call.  (a per-call hipMemcpyAsync from pageable memory is waited for on the host, which
    // serialised the enqueue of the backward launches with the execution of the forward pass and could not be stream-captured.)
    if (h->groups_dev.size() != h->groups_host.size() || memcmp(h->groups_dev.data(), h->groups_host.data(), ng * sizeof(LchGroup)) != 0) {
        if (hipStreamSynchronize(st) != hipSuccess) { err = "hipStreamSynchronize"; return MZL_E_HIP; }  // (an earlier step may still read the old table)
        if (hipMemcpy(h->d_groups, h->groups_host.data(), ng * sizeof(LchGroup), hipMemcpyHostToDevice) != hipSuccess) { err = "hipMemcpy"; return MZL_E_HIP; }
        h->groups_dev = h->groups_host;
    }
    LchArgs HA{};
    for (int i = 0; i < 3; i++) { HA.head[i] = h->head[i]; HA.lwT_off[i] = h->lwT_off[i]; }
    HA.groups = h->d_groups; HA.ngroups = ng; HA.K = K; HA.B = B; HA.P = h->P; HA.hw = h->hw; HA.A = h->A;
    HA.params = h->params; HA.grads = h->grads; HA.running = h->running; HA.nbt = h->nbt; HA.lwT = h->lwT;
    HA.u = h->hu; HA.dzb = h->hdz; HA.feat = h->hfeat; HA.dlogit = h->hdl; HA.spart = h->hspart; HA.spiv = h->hspiv; HA.coef = h->hcoef; HA.save = h->hsave; HA.lpart = h->hlpart;
    HA.n_max = h->n_max; HA.bt = bt; HA.loss = b->d_loss; HA.wpart = h->hwpart; HA.hp_total = h->hp_total;
    for (int i = 0; i < 3; i++) HA.hp_off[i] = h->hp_off[i];
    hipLaunchKernelGGL(k_lch_conv, dim3(B, ng), dim3(256), 0, st, HA);
    hipLaunchKernelGGL(k_lch_bn, dim3(1), dim3(64 * 3 * LCH_MAXOC), 0, st, HA);
    {
        const size_t lds = ((size_t)LCH_MAXOC * h->hw + 3 * (size_t)h->n_max + 32) * sizeof(float);
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
        ed.x = g_raw[t]; ed.gs = gs_next; ed.extra = h->dF_rew + tb; ed.partner = h->app_dyn[t].y.back(); ed.dz = h->defer_wgrad ? h->app_dyn[t].dz.back() : h->D[0][0]; ed.stat_part = h->stat[0];
        ed.scale = 0.5f; ed.B = B; ed.C = h->P; ed.hw = h->hw; ed.cpad = pad16(h->P);
        launch_entry(h, ed, st);
        LcEntry ep = ed;
        ep.x = f_out[t]; ep.gs = nullptr; ep.extra = h->dF_pred + tb; ep.partner = h->app_pred[t].y.back(); ep.dz = h->defer_wgrad ? h->app_pred[t].dz.back() : h->D[1][0]; ep.stat_part = h->stat[1];
        launch_entry(h, ep, st);
        float* gs_t = gs_bufs[t & 1];
        const int acc = t == K - 1 ? 0 : 1;
        ops.clear(); ops2.clear();
        s0.tower_bwd(ops, h->tower[1], h->app_dyn[t], h->s[t], h->act + (size_t)t * B, eg, acc, gs_t, h->GsP, h->defer_wgrad);
        s1.tower_bwd(ops2, h->tower[2], h->app_pred[t], h->s[t], nullptr, eg, acc, h->GsP, nullptr, h->defer_wgrad);
        if (run_zip(h, ops, ops2, false, st, paired)) { err = "internal: op lists of the paired towers do not line up"; return MZL_E_STATE; }
        gs_next = gs_t;
    }
    if (h->defer_wgrad) {  // the block layers of the two shared towers: K steps per launch, dynamics and prediction layer side by side
        ops.clear(); ops2.clear();
        const Sched* sc[2] = {&s0, &s1};
        std::vector<Op>* ol[2] = {&ops, &ops2};
        for (int tw = 0; tw < 2; tw++) {
            const TowerInfo& T = h->tower[1 + tw];
            const int li = T.conv0 ? 1 : 0, nl = (int)T.layers.size();
            for (int l = li; l < nl; l++)
                sc[tw]->wgrad_ops_steps(*ol[tw], h->layers[T.layers[l]], h->d_srcs + ((size_t)tw * h->srcs_stride[0] + l) * K, K, ((l - li) & 1) ? IN_BNRELU : IN_IDENT);
        }
        if (run_zip(h, ops, ops2, true, st, paired && ops.size() == ops2.size())) { err = "internal: op lists of the paired towers do not line up"; return MZL_E_STATE; }
    }
    if (h->atari) {
        atari_rep_bwd(h, B, gs_next, st);
    } else {
        LcEntry er{};
        er.x = hraw; er.gs = gs_next; er.extra = nullptr; er.partner = h->app_rep.y.back(); er.dz = h->D[0][0]; er.stat_part = h->stat[0];
        er.scale = 1.0f; er.B = B; er.C = h->P; er.hw = h->hw; er.cpad = pad16(h->P);
        launch_entry(h, er, st);
        ops.clear();
        sr.tower_bwd(ops, h->tower[0], h->app_rep, h->obs, nullptr, eg, 0, nullptr, nullptr);
        for (const Op& o : ops) launch_ops(h, &o, nullptr, st);
    }
    if (h->bad_dispatch) { err = "internal: a launch of the conv learner had no kernel build for its job (tap set / tile geometry)"; return MZL_E_INVALID; }
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

// diagnostic (tests): device pointers of saved tensors.  what: "y" (a = application: 0 representation, 1 + t dynamics_t, 1 + K + t prediction_t,
// 1 + 2 K the Atari net's res_blocks_3; b = layer of the tower), "x" (materialised post-ReLU outputs), "fcoef" (the layer's BatchNorm as applied:
// [3][pad16(C)], rows a, b), "s" (a = t), "dF_pred" / "dF_rew" (a = t).  Round 6 (tests/forced_masks.py: the decisions of THIS forward pass -- ReLU
// masks, normalisation arg-min / arg-max -- for a float64 reference that takes the same branches): "feat" the heads' post-ReLU features
// [3 K groups][B][2 hw]; the Atari representation's plane tensors "a1" / "a2" (post-ReLU conv_1 / conv_2), "s48_h1" / "s48_x" / "s24_h1" / "s24_x"
// (b = block: the block's inner post-ReLU tensor -- only with MZLC_KEEP_H1=1; else "s48_y1" / "s48_fcoef1" / .., its raw conv output and BatchNorm
// coefficients as for "y" / "fcoef" -- and the block's output), "hraw" (the pooled 6 x 6 state before normalisation).  *count: floats of the
// tensor at the last batch where the shape is known here, else the allocation's.
int mzlc_debug_tensor(const mzlc_learner* h, const char* what, int a, int b, void** ptr, int64_t* count) {
    const std::string w = what;
    const AppBufs* ap = nullptr;
    if (a == 0) ap = &h->app_rep;
    else if (a >= 1 && a <= h->K) ap = &h->app_dyn[a - 1];
    else if (a > h->K && a <= 2 * h->K) ap = &h->app_pred[a - 1 - h->K];
    else if (a == 2 * h->K + 1 && h->atari) ap = &h->a12;
    *count = (int64_t)h->T;
    if (ap == &h->a12) *count = (int64_t)h->maxB * h->P * (h->obsH / 8) * (h->obsW / 8);  // (the 12 x 12 stage of the Atari representation)
    if (w == "y" && ap && b >= 0 && b < (int)ap->y.size()) { *ptr = ap->y[b]; return MZL_OK; }
    if (w == "x" && ap && b >= 0 && b < (int)ap->x.size()) { *ptr = ap->x[b]; return MZL_OK; }
    if (w == "fcoef" && ap && b >= 0 && b < (int)ap->fcoef.size()) { *ptr = ap->fcoef[b]; *count = 3 * (int64_t)pad16(h->P); return MZL_OK; }
    if (w == "s" && a >= 0 && a < h->K) { *ptr = h->s[a]; return MZL_OK; }
    if (w == "feat") { *ptr = h->hfeat; *count = (int64_t)3 * h->K * h->lastB * LCH_MAXOC * h->hw; return MZL_OK; }
    if (h->atari) {
        const int64_t n48 = (int64_t)h->lastB * 128 * (h->obsH / 2) * (h->obsW / 2), n24 = (int64_t)h->lastB * h->P * (h->obsH / 4) * (h->obsW / 4);
        if (w == "a1") { *ptr = h->a1; *count = n48; return MZL_OK; }
        if (w == "a2") { *ptr = h->a2; *count = n24; return MZL_OK; }
        if (w == "hraw") { *ptr = h->hraw; *count = (int64_t)h->lastB * h->P * h->hw; return MZL_OK; }
        if (b >= 0 && b < 2) {
            if (w == "s48_y1") { *ptr = h->sb48.y[2 * b]; *count = n48; return MZL_OK; }
            if (w == "s24_y1") { *ptr = h->sb24.y[2 * b]; *count = n24; return MZL_OK; }
            if (w == "s48_fcoef1") { *ptr = h->sb48.fcoef[2 * b]; *count = 3 * 128; return MZL_OK; }
            if (w == "s24_fcoef1") { *ptr = h->sb24.fcoef[2 * b]; *count = 3 * (int64_t)pad16(h->P); return MZL_OK; }
            if (w == "s48_h1") { *ptr = h->sb48.h1[b]; *count = n48; return MZL_OK; }
            if (w == "s48_x") { *ptr = h->sb48.x[b]; *count = n48; return MZL_OK; }
            if (w == "s24_h1") { *ptr = h->sb24.h1[b]; *count = n24; return MZL_OK; }
            if (w == "s24_x") { *ptr = h->sb24.x[b]; *count = n24; return MZL_OK; }
        }
    }
    if (w == "gs") { *ptr = a ? h->GsB : h->GsA; return MZL_OK; }
    return MZL_E_INVALID;
}
