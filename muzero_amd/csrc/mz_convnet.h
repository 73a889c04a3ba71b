// mz_convnet.h -- host orchestration of the conv networks (board / Atari) and the HBM-resident search for them.
//
// MLP-class configs keep everything of a simulation inside one workgroup (mz_search_fast.h).  Conv-class configs do not
// fit: one Gomoku tree (A = 226, S = 200) is ~0.1 MB and one hidden state 115 KB.  Here a lock-step simulation is a short
// sequence of kernels over the whole env batch: tree select (HBM tree) -> dynamics tower -> reward head -> normalise ->
// prediction tower -> value head -> tree expand + backup.  The tree kernels are the SAME device functions as the
// LDS-resident search (mz_search.h: tree_select / tree_expand_backup / root_prior / tree_finish) pointed at a per-block
// global-memory region with the identical layout, so their parity with the reference carries over.
#pragma once
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "mz_conv.h"
#include "mz_search.h"
#include "mz_tower.h"

namespace mz {

struct HostTensorRef {
    const float* data;
    std::vector<int64_t> shape;
};
typedef std::map<std::string, HostTensorRef> ParamMap;

struct ConvLayerDev {
    float* w = nullptr;
    float* b = nullptr;
    int cin = 0, cin_real = 0, cout = 0, stride = 1;
};
struct ResBlockDev {
    ConvLayerDev c1, c2;
};
struct HeadDev {
    float *cw = nullptr, *cb = nullptr, *lw = nullptr, *lb = nullptr;
    int C = 0, oc = 0, n_out = 0;
};

struct ConvNetDev {
    int kind = 0;  // MZ_NET_BOARD (1) / MZ_NET_ATARI (2)
    int in_c = 0, in_h = 0, in_w = 0, A = 0, R = 0, P = 0, Sv = 1, Sr = 1, hh = 0, hw = 0;
    ConvLayerDev rep_conv, rep_conv2;
    std::vector<ResBlockDev> rep_res, dyn_res, pred_res;
    // fused-tower copies (mz_tower.h): the packed weights / biases of a tower's 2 * blocks convs back to back
    const float *tw_rep = nullptr, *tb_rep = nullptr, *tw_dyn = nullptr, *tb_dyn = nullptr, *tw_pred = nullptr, *tb_pred = nullptr;
    ConvLayerDev dyn_conv;
    // board games (gcd(h*w, A) == 1, planes % 16 == 0): the dynamics conv split into its real-channel part (dense MFMA kernel)
    // and the action channels' folded weights [cout][A][9] for k_action_sparse; dyn_inv_hw == 0: not available
    ConvLayerDev dyn_real;
    float* dyn_act_w = nullptr;
    int dyn_inv_hw = 0;
    // the same terms for the fused form (k_conv3x3's SP builds): weights transposed to [A * 9 + 1][cout] (last row zeros) and, per
    // (action, pixel), the byte offsets of the pixel's nine rows in chain order, 12 ints per pixel
    float* dyn_sp_w = nullptr;
    int* dyn_sp_terms = nullptr;
    HeadDev reward, policy, value;
    std::vector<void*> allocs;
    // work buffers (dense activations), sized by ensure_buffers
    float *bufA = nullptr, *bufB = nullptr, *bufC = nullptr;
    size_t buf_elems = 0;
    int hidden_size() const { return P * hh * hw; }
};

inline int env_int_early(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && v[0] ? atoi(v) : dflt;
}
// Diagnostic switches of the conv path (A/B measurements, tests; documented in include/mzplanner.h).  Read ONCE per process, at the first
// planner that needs them: a C-ABI caller gets one dispatch per process, never a different kernel from one call to the next.
struct ConvSwitches {
    int action_sparse;  // MZ_ACTION_SPARSE (1): the dynamics net's action planes as <= 9 folded weights per output (k_action_sparse / SP epilogue)
    int action_fuse;    // MZ_ACTION_FUSE (1): those terms in the first conv's epilogue instead of their own kernel
    int conv_spec;      // MZ_CONV_SPEC (1): the shape-specialised builds of k_conv3x3 / k_res_tower
    int tower;          // MZ_TOWER (1): a residual tower as one persistent kernel where the hidden state is small
    int conv_tile;      // MZ_CONV_TILE (0 = automatic): th * 100 + tw output tile of the tiled conv kernel
    int conv_g;         // MZ_CONV_G (-1 = automatic): images per workgroup
    int conv_nct;       // MZ_CONV_NCT (-1 = automatic): channel tiles per wave
};
inline const ConvSwitches& conv_switches() {
    static const ConvSwitches s = {env_int_early("MZ_ACTION_SPARSE", 1), env_int_early("MZ_ACTION_FUSE", 1), env_int_early("MZ_CONV_SPEC", 1),
                                   env_int_early("MZ_TOWER", 1), env_int_early("MZ_CONV_TILE", 0), env_int_early("MZ_CONV_G", -1), env_int_early("MZ_CONV_NCT", -1)};
    return s;
}

inline hipError_t dev_upload(ConvNetDev& n, const std::vector<float>& h, float** d) {
    hipError_t e = hipMalloc(d, h.size() * sizeof(float));
    if (e != hipSuccess) return e;
    n.allocs.push_back(*d);
    return hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
}

inline hipError_t dev_upload(ConvNetDev& n, const std::vector<int>& h, int** d) {
    float* f = nullptr;
    hipError_t e = hipMalloc(&f, h.size() * sizeof(int));
    if (e != hipSuccess) return e;
    n.allocs.push_back(f);
    *d = reinterpret_cast<int*>(f);
    return hipMemcpy(f, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
}

// Conv (bias-free) + optional eval-mode BatchNorm folded exactly like the oracle's conv_init: alpha = (1/sqrt(var+eps))*gamma,
// w' = w*alpha, b' = beta - mean*alpha (float32 operations in that order); weights packed into the MFMA A-fragment order
// [co_tile][cb][tap][lane][4]:  W'[16t + (lane&15)][cb*16 + 4s + (lane>>4)][tap]
inline int build_conv(ConvNetDev& n, const ParamMap& pm, const std::string& conv_w, const std::string& bn, int cin, int cin_real, int cout,
                      int k, int stride, ConvLayerDev* out, std::string* err, int pack_cin = -1, float** tail_w = nullptr, float** tail_wT = nullptr) {
    auto wi = pm.find(conv_w);
    if (wi == pm.end()) { *err = "missing parameter " + conv_w; return -1; }
    const HostTensorRef& W = wi->second;
    if (W.shape.size() != 4 || W.shape[0] != cout || W.shape[1] != cin || W.shape[2] != k || W.shape[3] != k) {
        *err = "shape mismatch for " + conv_w;
        return -1;
    }
    std::vector<float> alpha(cout, 1.0f), bias(((cout + 15) / 16) * 16, 0.0f);
    if (!bn.empty()) {
        const char* names[4] = {".weight", ".bias", ".running_mean", ".running_var"};
        const float* t[4];
        for (int i = 0; i < 4; i++) {
            auto it = pm.find(bn + names[i]);
            if (it == pm.end() || it->second.shape.size() != 1 || it->second.shape[0] != cout) { *err = "missing/mis-shaped " + bn + names[i]; return -1; }
            t[i] = it->second.data;
        }
        for (int co = 0; co < cout; co++) {
            const float invstd = 1.0f / sqrtf(t[3][co] + 1e-5f);
            alpha[co] = invstd * t[0][co];
            const float m = t[2][co] * alpha[co];
            bias[co] = t[1][co] - m;
        }
    }
    // pack_cin >= 0: pack only the first pack_cin input channels (and hand the folded weights of the rest to *tail_w as
    // [cout][cin - pack_cin][taps])
    const int pcin = pack_cin >= 0 ? pack_cin : cin;
    const int taps = k * k, n_cb = (pcin + 15) / 16, co_tiles = (cout + 15) / 16;
    if (tail_w) {
        const int nt = cin - pcin;
        std::vector<float> tw((size_t)cout * nt * taps);
        for (int co = 0; co < cout; co++)
            for (int c = 0; c < nt; c++)
                for (int tap = 0; tap < taps; tap++) {
                    float v = W.data[((size_t)co * cin + pcin + c) * taps + tap];
                    if (!bn.empty()) v = v * alpha[co];
                    tw[((size_t)co * nt + c) * taps + tap] = v;
                }
        if (dev_upload(n, tw, tail_w) != hipSuccess) { *err = "hipMalloc/hipMemcpy failed"; return -2; }
        if (tail_wT) {  // [nt * taps + 1][cout], row c * taps + tap; the last row stays zero
            std::vector<float> twt((size_t)(nt * taps + 1) * cout, 0.0f);
            for (int co = 0; co < cout; co++)
                for (int c = 0; c < nt; c++)
                    for (int tap = 0; tap < taps; tap++) twt[((size_t)c * taps + tap) * cout + co] = tw[((size_t)co * nt + c) * taps + tap];
            if (dev_upload(n, twt, tail_wT) != hipSuccess) { *err = "hipMalloc/hipMemcpy failed"; return -2; }
        }
    }
    std::vector<float> pw((size_t)co_tiles * n_cb * taps * 256, 0.0f);
    for (int t = 0; t < co_tiles; t++)
        for (int cb = 0; cb < n_cb; cb++)
            for (int tap = 0; tap < taps; tap++)
                for (int lane = 0; lane < 64; lane++)
                    for (int s = 0; s < 4; s++) {
                        const int co = 16 * t + (lane & 15), ci = cb * 16 + 4 * s + (lane >> 4);
                        if (co < cout && ci < pcin) {
                            float v = W.data[((size_t)co * cin + ci) * taps + tap];
                            if (!bn.empty()) v = v * alpha[co];
                            pw[((((size_t)t * n_cb + cb) * taps + tap) * 64 + lane) * 4 + s] = v;
                        }
                    }
    out->cin = pcin; out->cin_real = pack_cin >= 0 ? pcin : cin_real; out->cout = cout; out->stride = stride;
    if (dev_upload(n, pw, &out->w) != hipSuccess || dev_upload(n, bias, &out->b) != hipSuccess) { *err = "hipMalloc/hipMemcpy failed"; return -2; }
    return 0;
}

inline int build_res(ConvNetDev& n, const ParamMap& pm, const std::string& prefix, int planes, ResBlockDev* r, std::string* err) {
    int rc = build_conv(n, pm, prefix + ".conv_block1.0.weight", prefix + ".conv_block1.1", planes, planes, planes, 3, 1, &r->c1, err);
    if (rc) return rc;
    return build_conv(n, pm, prefix + ".conv_block2.0.weight", prefix + ".conv_block2.1", planes, planes, planes, 3, 1, &r->c2, err);
}

// head: 1x1 conv + BN (folded, plain [oc][C] layout) + Linear
inline int build_head(ConvNetDev& n, const ParamMap& pm, const std::string& prefix, int C, int oc, int hw, int n_out, HeadDev* h, std::string* err) {
    auto wi = pm.find(prefix + ".0.weight");
    auto lw = pm.find(prefix + ".4.weight"), lb = pm.find(prefix + ".4.bias");
    if (wi == pm.end() || lw == pm.end() || lb == pm.end()) { *err = "missing head parameters " + prefix; return -1; }
    const char* names[4] = {".1.weight", ".1.bias", ".1.running_mean", ".1.running_var"};
    const float* t[4];
    for (int i = 0; i < 4; i++) {
        auto it = pm.find(prefix + names[i]);
        if (it == pm.end()) { *err = "missing " + prefix + names[i]; return -1; }
        t[i] = it->second.data;
    }
    if (lw->second.shape.size() != 2 || lw->second.shape[0] != n_out || lw->second.shape[1] != (int64_t)oc * hw) { *err = "shape mismatch for " + prefix + ".4.weight"; return -1; }
    std::vector<float> cw((size_t)oc * C), cb(oc);
    for (int o = 0; o < oc; o++) {
        const float invstd = 1.0f / sqrtf(t[3][o] + 1e-5f);
        const float alpha = invstd * t[0][o];
        const float m = t[2][o] * alpha;
        cb[o] = t[1][o] - m;
        for (int c = 0; c < C; c++) cw[(size_t)o * C + c] = wi->second.data[(size_t)o * C + c] * alpha;
    }
    std::vector<float> l(lw->second.data, lw->second.data + (size_t)n_out * oc * hw), b(lb->second.data, lb->second.data + n_out);
    h->C = C; h->oc = oc; h->n_out = n_out;
    if (dev_upload(n, cw, &h->cw) != hipSuccess || dev_upload(n, cb, &h->cb) != hipSuccess || dev_upload(n, l, &h->lw) != hipSuccess ||
        dev_upload(n, b, &h->lb) != hipSuccess) { *err = "hipMalloc/hipMemcpy failed"; return -2; }
    return 0;
}

inline void convnet_free(ConvNetDev& n) {
    for (void* p : n.allocs) (void)hipFree(p);
    n.allocs.clear();
    for (float* p : {n.bufA, n.bufB, n.bufC})
        if (p) (void)hipFree(p);
    n.bufA = n.bufB = n.bufC = nullptr;
    n.buf_elems = 0;
    n.rep_res.clear(); n.dyn_res.clear(); n.pred_res.clear();
}

inline int convnet_build(ConvNetDev& n, const ParamMap& pm, std::string* err) {
    int rc;
    const int P = n.P, R = n.R;
    if (n.kind == 1) {  // MuZeroBoardGameNet, network.py:540-574
        n.hh = n.in_h; n.hw = n.in_w;
        if ((rc = build_conv(n, pm, "represent_net.conv_block.0.weight", "represent_net.conv_block.1", n.in_c, n.in_c, P, 3, 1, &n.rep_conv, err))) return rc;
        n.rep_res.resize(R);
        for (int i = 0; i < R; i++)
            if ((rc = build_res(n, pm, "represent_net.res_blocks." + std::to_string(i), P, &n.rep_res[i], err))) return rc;
    } else {  // MuZeroAtariNet, network.py:501-537: conv_1 always has 128 output channels (network.py:324)
        n.hh = 6; n.hw = 6;
        if ((rc = build_conv(n, pm, "represent_net.conv_1.weight", "", n.in_c, n.in_c, 128, 3, 2, &n.rep_conv, err))) return rc;
        n.rep_res.resize(6);
        for (int i = 0; i < 2; i++)
            if ((rc = build_res(n, pm, "represent_net.res_blocks_1." + std::to_string(i), 128, &n.rep_res[i], err))) return rc;
        if ((rc = build_conv(n, pm, "represent_net.conv_2.weight", "", 128, 128, P, 3, 2, &n.rep_conv2, err))) return rc;
        for (int i = 0; i < 2; i++)
            if ((rc = build_res(n, pm, "represent_net.res_blocks_2." + std::to_string(i), P, &n.rep_res[2 + i], err))) return rc;
        for (int i = 0; i < 2; i++)
            if ((rc = build_res(n, pm, "represent_net.res_blocks_3." + std::to_string(i), P, &n.rep_res[4 + i], err))) return rc;
    }
    const int hw = n.hh * n.hw;
    if ((rc = build_conv(n, pm, "dynamics_net.conv_block.0.weight", "dynamics_net.conv_block.1", P + n.A, P, P, 3, 1, &n.dyn_conv, err))) return rc;
    {
        auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
        n.dyn_inv_hw = 0;
        if ((P & 15) == 0 && gcd(hw, n.A) == 1 && conv_switches().action_sparse) {
            if ((rc = build_conv(n, pm, "dynamics_net.conv_block.0.weight", "dynamics_net.conv_block.1", P + n.A, P, P, 3, 1, &n.dyn_real, err, P,
                                 &n.dyn_act_w, &n.dyn_sp_w)))
                return rc;
            for (int i = 1; i < n.A; i++)
                if ((long long)i * (hw % n.A) % n.A == 1) n.dyn_inv_hw = i;
            if (n.A == 1) n.dyn_inv_hw = 0;
            if (n.dyn_inv_hw > 0) {
                // k_action_sparse's term list (key = chain position: 16-channel block, tap, channel in block), for every (action, pixel)
                std::vector<int> terms((size_t)n.A * hw * 12, 0);
                const int zero_row = n.A * 9 * P * (int)sizeof(float);
                for (int a = 0; a < n.A; a++)
                    for (int p = 0; p < hw; p++) {
                        int key[9], cnt = 0;
                        for (int t = 0; t < 9; t++) {
                            const int y = p / n.hw + t / 3 - 1, x = p % n.hw + t % 3 - 1;
                            if (y < 0 || y >= n.hh || x < 0 || x >= n.hw) continue;
                            int d = (a - (y * n.hw + x)) % n.A;
                            d = d < 0 ? d + n.A : d;
                            const int c = (int)(((long long)d * n.dyn_inv_hw) % n.A);
                            key[cnt++] = ((c >> 4) * 9 + t) * 16 + (c & 15);
                        }
                        std::sort(key, key + cnt);
                        int* row = &terms[((size_t)a * hw + p) * 12];
                        for (int t = 0; t < 12; t++) row[t] = zero_row;
                        for (int t = 0; t < cnt; t++) {
                            const int k = key[t], c = (k / 144) * 16 + (k & 15), tap = (k % 144) / 16;
                            row[t] = (c * 9 + tap) * P * (int)sizeof(float);
                        }
                    }
                if (dev_upload(n, terms, &n.dyn_sp_terms) != hipSuccess) { *err = "hipMalloc/hipMemcpy failed"; return -2; }
            }
        }
    }
    n.dyn_res.resize(R);
    n.pred_res.resize(R);
    for (int i = 0; i < R; i++) {
        if ((rc = build_res(n, pm, "dynamics_net.res_blocks." + std::to_string(i), P, &n.dyn_res[i], err))) return rc;
        if ((rc = build_res(n, pm, "prediction_net.res_blocks." + std::to_string(i), P, &n.pred_res[i], err))) return rc;
    }
    if ((rc = build_head(n, pm, "dynamics_net.reward_head", P, 1, hw, n.Sr, &n.reward, err))) return rc;
    if ((rc = build_head(n, pm, "prediction_net.policy_net", P, 2, hw, n.A, &n.policy, err))) return rc;
    if ((rc = build_head(n, pm, "prediction_net.value_net", P, 1, hw, n.Sv, &n.value, err))) return rc;
    auto table = [&](const std::vector<ResBlockDev>& blocks, const float** w, const float** b) -> int {
        if (blocks.empty() || P > 128 || (P & 15)) return 0;
        const size_t wn = (size_t)(P / 16) * (P / 16) * 9 * 256, n_convs = 2 * blocks.size();
        float *dw = nullptr, *db = nullptr;
        if (hipMalloc(&dw, n_convs * wn * sizeof(float)) != hipSuccess || hipMalloc(&db, n_convs * P * sizeof(float)) != hipSuccess) return -2;
        n.allocs.push_back(dw); n.allocs.push_back(db);
        size_t i = 0;
        for (const ResBlockDev& r : blocks)
            for (const ConvLayerDev* c : {&r.c1, &r.c2}) {
                if (c->cin != P || c->cout != P) return 0;  // (Atari's 128-plane stage with another P: separate launches)
                if (hipMemcpy(dw + i * wn, c->w, wn * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) return -2;
                if (hipMemcpy(db + i * P, c->b, P * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) return -2;
                i++;
            }
        *w = dw; *b = db;
        return 0;
    };
    if (n.kind == 1 && table(n.rep_res, &n.tw_rep, &n.tb_rep)) { *err = "hipMalloc/hipMemcpy failed"; return -2; }
    if (table(n.dyn_res, &n.tw_dyn, &n.tb_dyn) || table(n.pred_res, &n.tw_pred, &n.tb_pred)) { *err = "hipMalloc/hipMemcpy failed"; return -2; }
    return 0;
}

inline hipError_t convnet_ensure_buffers(ConvNetDev& n, int B) {
    size_t per = (size_t)n.P * n.hh * n.hw;
    if (n.kind == 2) {
        const size_t a = (size_t)128 * ((n.in_h + 1) / 2) * ((n.in_w + 1) / 2);
        per = a > per ? a : per;
    }
    const size_t need = per * B;
    if (need * sizeof(float) >= ((size_t)1 << 32)) return hipErrorInvalidValue;  // 32-bit buffer offsets in k_conv3x3: split the batch
    if (need <= n.buf_elems) return hipSuccess;
    for (float** p : {&n.bufA, &n.bufB, &n.bufC}) {
        if (*p) (void)hipFree(*p);
        hipError_t e = hipMalloc(p, (need + 64) * sizeof(float));  // + slack: the last pixel quad of a channel is read as 16 bytes
        if (e != hipSuccess) return e;
    }
    n.buf_elems = need;
    return hipSuccess;
}

// Launch geometry of one 3x3 conv: spatial tile, images per workgroup (small boards share a workgroup so that the pixel
// dimension fills whole 16-wide MFMA tiles), pixel tiles per wave (NPT) and channel tiles per wave (NCT).
struct ConvGeom {
    int th, tw, G, npt, nct, cstride, qstride;
    bool whole;  // tile == whole image, stride 1: quad-based staging (k_conv3x3<.., true>)
};


inline ConvGeom conv_geometry(int B, int oh, int ow, int stride, int cout, bool allow_group = true) {
    static const int kNpt[9] = {1, 2, 3, 4, 5, 6, 9, 12, 15};
    auto round_npt = [&](int tiles) {
        for (int c : kNpt)
            if (tiles <= c) return c;
        return 0;
    };
    ConvGeom g{};
    const bool whole = stride == 1 && oh * ow <= 240 && (oh + 2) * (ow + 2) <= 384;
    const int QP = (oh * ow + 3) / 4;  // pixel quads per image: one lane each
    // tiled images: 8x8 output tiles (4 pixel tiles, a 10x10 slab), or, where the image divides and the stride-1 slab fits, 12x16 /
    // 12x12 (12 / 9 pixel tiles: 1.31 / 1.36 staged positions per output pixel instead of 1.56, and 3x / 2.25x the MFMAs behind every
    // barrier; C4 +0.6 %).  MZ_CONV_TILE = th * 100 + tw overrides (diagnostics).
    int tth = 8, ttw = 8;
    if (!whole && stride == 1) {
        const int forced = conv_switches().conv_tile;
        const int cand[3][2] = {{forced / 100, forced % 100}, {12, 16}, {12, 12}};
        for (int i = forced ? 0 : 1; i < 3; i++) {
            const int wh = cand[i][0], ww = cand[i][1];
            if (wh < 1 || ww < 1 || oh % wh || ow % ww || wh * ww > 240 || (wh + 2) * (ww + 2) > 384) continue;
            const int n = round_npt((wh * ww + 15) / 16);
            if (n && (n <= 4 || n >= 9)) { tth = wh; ttw = ww; break; }  // (the tiled kernel is instantiated for 1-4, 9, 12 and 15 pixel tiles)
        }
    }
    g.th = whole ? oh : tth; g.tw = whole ? ow : ttw;
    const int plane = ((g.th - 1) * stride + 3) * ((g.tw - 1) * stride + 3), TP = g.th * g.tw;
    const int zs1 = (cout + 63) / 64;  // channel slices with NCT = 1
    g.G = 1;
    if (whole && allow_group) {
        // best MFMA fill among group sizes that keep at least ~2 workgroups per CU busy (when the batch allows it)
        double best = -1.0;
        for (int G = 1; G <= 16 && G <= B && G * plane <= 384 && G * QP <= 64; G++) {
            const int npt = round_npt((G * TP + 15) / 16);
            if (!npt || (G > 1 && npt > 9)) continue;
            const long wgs = (long)((B + G - 1) / G) * zs1;
            double fill = (double)(G * TP) / (16.0 * npt);
            if (wgs < 256) fill *= (double)wgs / 256.0;
            if (fill > best + 1e-9) { best = fill; g.G = G; }
        }
    }
    if (conv_switches().conv_g >= 0) g.G = conv_switches().conv_g;
    g.whole = whole;
    g.npt = round_npt((g.G * TP + 15) / 16);
    const long tiles = (long)((oh + g.th - 1) / g.th) * ((ow + g.tw - 1) / g.tw);
    const long wgs1 = tiles * ((B + g.G - 1) / g.G) * zs1;
    g.nct = (cout > 64 && wgs1 >= 1024) ? 2 : 1;  // two channel tiles per wave halve the staging work when there are workgroups to spare
    if (conv_switches().conv_nct >= 0) g.nct = conv_switches().conv_nct;
    g.qstride = (g.G * plane * 4 + 63) & ~63;
    g.cstride = 4 * g.qstride;
    return g;
}

template <int NCT, bool WHOLE, bool SP = false>
inline void conv_launch_npt(int npt, dim3 grid, size_t lds, hipStream_t st, const ConvLaunch& L) {
    const dim3 block(256);
    switch (npt) {
        case 1: hipLaunchKernelGGL((k_conv3x3<1, NCT, WHOLE, 0, SP>), grid, block, lds, st, L); break;
        case 2: hipLaunchKernelGGL((k_conv3x3<2, NCT, WHOLE, 0, SP>), grid, block, lds, st, L); break;
        case 3: hipLaunchKernelGGL((k_conv3x3<3, NCT, WHOLE, 0, SP>), grid, block, lds, st, L); break;
        case 4: hipLaunchKernelGGL((k_conv3x3<4, NCT, WHOLE, 0, SP>), grid, block, lds, st, L); break;
        default:
            if constexpr (WHOLE) {
                switch (npt) {
                    case 5: hipLaunchKernelGGL((k_conv3x3<5, NCT, true, 0, SP>), grid, block, lds, st, L); break;
                    case 6: hipLaunchKernelGGL((k_conv3x3<6, NCT, true, 0, SP>), grid, block, lds, st, L); break;
                    case 9: hipLaunchKernelGGL((k_conv3x3<9, NCT, true, 0, SP>), grid, block, lds, st, L); break;
                    case 12: hipLaunchKernelGGL((k_conv3x3<12, NCT, true, 0, SP>), grid, block, lds, st, L); break;
                    default: hipLaunchKernelGGL((k_conv3x3<15, NCT, true, 0, SP>), grid, block, lds, st, L); break;
                }
            }
            else if (npt == 9) hipLaunchKernelGGL((k_conv3x3<9, NCT, false>), grid, block, lds, st, L);  // 12x12 tiles
            else if (npt == 12) hipLaunchKernelGGL((k_conv3x3<12, NCT, false>), grid, block, lds, st, L);
            else if (npt == 15) hipLaunchKernelGGL((k_conv3x3<15, NCT, false>), grid, block, lds, st, L);
            break;  // tiled images use 8x8 tiles (npt == 4) or 12x12 (npt == 9)
    }
}

static long long* g_conv_stamps = nullptr;  // diagnostic builds only (tools/micro/conv_bench.hip)

// one 3x3 conv launch; input either dense `in` or per-image `in_ptrs`
// (gathered input: in_base / in_span_floats describe the store the row pointers point into)
// sparse action terms for the SP builds (null: none); conv_run returns whether it fused them (the caller runs k_action_sparse otherwise)
struct ConvSparse {
    const float* w;
    const int* terms;
    const int* action;
};

inline bool conv_run(hipStream_t st, const ConvLayerDev& Lr, int B, const float* in, const float* const* in_ptrs, const int* action, int A, int ih,
                     int iw, const float* residual, float* out, bool relu, const float* in_base = nullptr, size_t in_span_floats = 0,
                     const ConvSparse* sp = nullptr) {
    ConvLaunch L{};
    L.in_ptrs = in_ptrs; L.in = in; L.in_base = in_base; L.action = action; L.num_actions = A > 0 ? A : 1;
    L.cin_real = Lr.cin_real; L.cin = Lr.cin; L.ih = ih; L.iw = iw; L.stride = Lr.stride;
    L.oh = (ih + 2 - 3) / Lr.stride + 1; L.ow = (iw + 2 - 3) / Lr.stride + 1;
    L.cout = Lr.cout; L.w = Lr.w; L.bias = Lr.b; L.residual = residual; L.out = out; L.relu = relu ? 1 : 0; L.B = B;
    // images of one workgroup share a buffer descriptor: their rows must lie within 32-bit byte offsets of the store base
    const bool group_ok = !in_ptrs || (in_base && in_span_floats < ((size_t)1 << 30));
    const ConvGeom g = conv_geometry(B, L.oh, L.ow, Lr.stride, Lr.cout, group_ok);
    L.th = g.th; L.tw = g.tw; L.G = g.G; L.cstride = g.cstride; L.qstride = g.qstride; L.stamps = g_conv_stamps;
    L.tiles_x = (L.ow + L.tw - 1) / L.tw; L.tiles_y = (L.oh + L.th - 1) / L.th;
    size_t lds = (size_t)2 * g.cstride * sizeof(float);
    const dim3 grid(L.tiles_x * L.tiles_y, (B + g.G - 1) / g.G, (Lr.cout + 64 * g.nct - 1) / (64 * g.nct));
    // SP: whole-image builds with one 64-channel slice per workgroup, stride 1; the term rows of the workgroup's images live behind the slabs
    const bool sp_on = conv_switches().action_fuse != 0;
    const bool fuse = sp && sp_on && g.whole && g.nct == 1 && L.stride == 1 && L.oh == L.ih && L.ow == L.iw &&
                      lds + (size_t)g.G * ih * iw * 48 <= 64 * 1024;
    if (fuse) {
        L.sp_w = sp->w; L.sp_terms = sp->terms; L.sp_action = sp->action;
        lds += (size_t)g.G * ih * iw * 48;
    } else if (sp) {
        L.relu = 0;  // k_action_sparse reads the pre-activation and applies the ReLU after its additions
    }
    // shape-specialised builds (mz_conv.h, SIDE): a whole 15x15 image per workgroup, 128 output channels (Gomoku's towers)
    const bool spec_ok = conv_switches().conv_spec != 0;
    if (spec_ok && g.whole && g.nct == 1 && g.npt == 15 && g.G == 1 && L.stride == 1 && L.ih == 15 && L.iw == 15 && L.oh == 15 && L.ow == 15 &&
        Lr.cout == 128 && L.tiles_x == 1 && L.tiles_y == 1 && g.qstride == (((15 + 2) * (15 + 2) * 4 + 63) & ~63) && g.cstride == 4 * g.qstride) {
        if (fuse) hipLaunchKernelGGL((k_conv3x3<15, 1, true, 15, true>), grid, dim3(256), lds, st, L);
        else hipLaunchKernelGGL((k_conv3x3<15, 1, true, 15>), grid, dim3(256), lds, st, L);
        return fuse;
    }
    if (spec_ok && !g.whole && g.nct == 2 && g.G == 1 && L.stride == 1 && Lr.cout == 128 && L.ih == L.oh && L.iw == L.ow && L.oh == L.ow && g.th == 12 &&
        L.tiles_x == L.ow / g.tw && L.tiles_y == L.oh / 12 && g.qstride == ((14 * (g.tw + 2) * 4 + 63) & ~63) && g.cstride == 4 * g.qstride) {
        // the Atari representation's 48 x 48 (12 x 16 tiles) and 24 x 24 (12 x 12 tiles) layers
        if (L.oh == 48 && g.tw == 16 && g.npt == 12) { hipLaunchKernelGGL((k_conv3x3<12, 2, false, 48>), grid, dim3(256), lds, st, L); return false; }
        if (L.oh == 24 && g.tw == 12 && g.npt == 9) { hipLaunchKernelGGL((k_conv3x3<9, 2, false, 24>), grid, dim3(256), lds, st, L); return false; }
    }
    if (g.whole) {
        if (fuse) conv_launch_npt<1, true, true>(g.npt, grid, lds, st, L);
        else if (g.nct == 2) conv_launch_npt<2, true>(g.npt, grid, lds, st, L);
        else conv_launch_npt<1, true>(g.npt, grid, lds, st, L);
    } else {
        if (g.nct == 2) conv_launch_npt<2, false>(g.npt, grid, lds, st, L);
        else conv_launch_npt<1, false>(g.npt, grid, lds, st, L);
    }
    return fuse;
}

// Geometry of the fused LDS-resident tower (mz_tower.h) for a P-channel tower on h x w images, or npt == 0 if it does not apply
struct TowerGeom {
    int G, npt, nposp;
    size_t lds;
};

inline TowerGeom tower_geometry(int B, int P, int h, int w) {
    TowerGeom best{0, 0, 0, 0};
    if (P > 128 || (P & 15) || conv_switches().tower == 0) return best;
    const int hw = h * w, n_cb = P / 16;
    double best_score = -1.0;
    for (int G = 1; G <= 16 && G <= B; G++) {
        const int npt = (G * hw + 15) / 16;
        if (npt > 6) break;
        const int nposp = npt * 16 + ((G * hw) % 16 == 0 ? 16 : 0);  // at least one padding position (it stays zero: the halo reads)
        const size_t lds = (size_t)3 * n_cb * 4 * nposp * 4 * sizeof(float);
        if (lds > 156 * 1024) break;
        const long wgs = (B + G - 1) / G;
        double score = (double)(G * hw) / (16.0 * npt);
        if (wgs < 256) score *= (double)wgs / 256.0;
        if (score > best_score + 1e-9) { best_score = score; best = TowerGeom{G, npt, nposp, lds}; }
    }
    return best;
}

template <int NPT>
inline void tower_launch_npt(hipStream_t st, const TowerLaunch& L, int wgs, size_t lds) {
    static bool attr_set = false;  // one planner thread per process configures this instantiation once
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_res_tower<NPT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_res_tower<NPT>, dim3(wgs), dim3(512), lds, st, L);
}

// residual tower on x (dense), t1/t2 scratch; returns the buffer holding the result.  Small images take the fused
// LDS-resident kernel (tables tw / tb: per-conv weight / bias pointers of `blocks`), the rest one k_conv3x3 launch per conv.
inline float* tower_run(hipStream_t st, const std::vector<ResBlockDev>& blocks, int first, int count, int B, float* x, float* t1, float* t2, int h, int w,
                        const float* tw = nullptr, const float* tb = nullptr) {
    if (count > 0 && tw && tb) {
        const TowerGeom g = tower_geometry(B, blocks[first].c1.cout, h, w);
        if (g.npt) {
            TowerLaunch L{};
            const int Pc = blocks[first].c1.cout;
            L.in = x; L.out = t1; L.w = tw + (size_t)2 * first * (Pc / 16) * (Pc / 16) * 9 * 256; L.bias = tb + 2 * first * Pc; L.n_convs = 2 * count;
            L.P = Pc; L.h = h; L.w_img = w; L.G = g.G; L.B = B; L.nposp = g.nposp; L.stamps = nullptr;
            const int wgs = (B + g.G - 1) / g.G;
            switch (g.npt) {
                case 1: tower_launch_npt<1>(st, L, wgs, g.lds); break;
                case 2: tower_launch_npt<2>(st, L, wgs, g.lds); break;
                case 3: tower_launch_npt<3>(st, L, wgs, g.lds); break;
                case 4: tower_launch_npt<4>(st, L, wgs, g.lds); break;
                case 5:
                    if (conv_switches().conv_spec != 0 && Pc == 128 && h == 6 && w == 6 && g.G == 2) {  // (mz_tower.h, SPEC == 1)
                        static bool attr1 = false;
                        if (!attr1) {
                            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_res_tower<5, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                            attr1 = true;
                        }
                        hipLaunchKernelGGL((k_res_tower<5, 1>), dim3(wgs), dim3(512), g.lds, st, L);
                    } else tower_launch_npt<5>(st, L, wgs, g.lds);
                    break;
                default: tower_launch_npt<6>(st, L, wgs, g.lds); break;
            }
            return t1;
        }
    }
    for (int i = first; i < first + count; i++) {
        conv_run(st, blocks[i].c1, B, x, nullptr, nullptr, 0, h, w, nullptr, t1, true);
        conv_run(st, blocks[i].c2, B, t1, nullptr, nullptr, 0, h, w, x, t2, true);
        float* s = x; x = t2; t2 = s;
    }
    return x;
}

inline void head_run(hipStream_t st, const HeadDev& H, int B, const float* in, const float* const* in_ptrs, int hw, int mode, float* out_scalar,
                     float* out_probs) {
    HeadLaunch L{};
    L.in = in; L.in_ptrs = in_ptrs; L.C = H.C; L.hw = hw; L.oc = H.oc; L.n_out = H.n_out; L.cw = H.cw; L.cb = H.cb; L.lw = H.lw; L.lb = H.lb;
    L.mode = mode; L.out_scalar = out_scalar; L.out_probs = out_probs; L.B = B;
    const size_t lds = ((((size_t)H.oc * hw + 3) & ~(size_t)3) + (((size_t)H.n_out + 3) & ~(size_t)3) + (size_t)HEAD_CK * hw) * sizeof(float);
    hipLaunchKernelGGL(k_head, dim3(B), dim3(256), lds, st, L);
}

// the two of {a, b, c} that are not x
inline void other_two(float* a, float* b, float* c, const float* x, float** o1, float** o2) {
    float* r[2];
    int k = 0;
    for (float* p : {a, b, c})
        if (p != x && k < 2) r[k++] = p;
    *o1 = r[0]; *o2 = r[1];
}

// shared tail of network.py:62-84 and :86-111: normalise x (dense) into the node-store rows / dense destination, run the
// prediction tower and the heads on the normalised state
inline void convnet_tail(hipStream_t st, ConvNetDev& n, int B, float* x, float* const* dst_ptrs, float* dst_dense, float* pi, float* value) {
    const int hw = n.hh * n.hw;
    float *nrm, *t2;
    other_two(n.bufA, n.bufB, n.bufC, x, &nrm, &t2);
    {
        const dim3 grid((hw + 31) / 32, B), block(256);
        const int cpt = (n.P + 7) / 8;
        if (cpt <= 2) hipLaunchKernelGGL(k_normalize_planes<2>, grid, block, 0, st, x, dst_ptrs, dst_dense, nrm, B, n.P, hw);
        else if (cpt <= 8) hipLaunchKernelGGL(k_normalize_planes<8>, grid, block, 0, st, x, dst_ptrs, dst_dense, nrm, B, n.P, hw);
        else if (cpt <= 16) hipLaunchKernelGGL(k_normalize_planes<16>, grid, block, 0, st, x, dst_ptrs, dst_dense, nrm, B, n.P, hw);
        else if (cpt <= 32) hipLaunchKernelGGL(k_normalize_planes<32>, grid, block, 0, st, x, dst_ptrs, dst_dense, nrm, B, n.P, hw);
        else hipLaunchKernelGGL(k_normalize_planes<64>, grid, block, 0, st, x, dst_ptrs, dst_dense, nrm, B, n.P, hw);
    }
    float* f = tower_run(st, n.pred_res, 0, n.R, B, nrm, x, t2, n.hh, n.hw, n.tw_pred, n.tb_pred);
    if (pi) head_run(st, n.policy, B, f, nullptr, hw, 1, nullptr, pi);
    head_run(st, n.value, B, f, nullptr, hw, 0, value, nullptr);
}

// network.py:62-84: obs dense [B][c][h][w] -> normalised hidden rows (dst_ptrs[b] and/or dense dst), pi0 [B][A], value [B]
inline void convnet_initial(hipStream_t st, ConvNetDev& n, int B, const float* obs, float* const* dst_ptrs, float* dst_dense, float* pi,
                            float* value) {
    float *a = n.bufA, *b = n.bufB, *c = n.bufC, *o1, *o2;
    int h = n.in_h, w = n.in_w;
    float* x;
    conv_run(st, n.rep_conv, B, obs, nullptr, nullptr, 0, h, w, nullptr, a, true);  // board: conv+BN+ReLU (:389); Atari: conv_1+ReLU (:346)
    if (n.kind == 1) {
        x = tower_run(st, n.rep_res, 0, n.R, B, a, b, c, h, w, n.tw_rep, n.tb_rep);
    } else {
        h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1;
        x = tower_run(st, n.rep_res, 0, 2, B, a, b, c, h, w);
        other_two(a, b, c, x, &o1, &o2);
        conv_run(st, n.rep_conv2, B, x, nullptr, nullptr, 0, h, w, nullptr, o1, true);  // conv_2 + ReLU (:348)
        h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1;
        x = tower_run(st, n.rep_res, 2, 2, B, o1, o2, x, h, w);
        other_two(a, b, c, x, &o1, &o2);
        int h2 = (h - 1) / 2 + 1, w2 = (w - 1) / 2 + 1;
        hipLaunchKernelGGL(k_avgpool, dim3(1024), dim3(256), 0, st, x, o1, B, n.P, h, w, h2, w2);
        h = h2; w = w2;
        x = tower_run(st, n.rep_res, 4, 2, B, o1, o2, x, h, w);
        other_two(a, b, c, x, &o1, &o2);
        h2 = (h - 1) / 2 + 1; w2 = (w - 1) / 2 + 1;
        hipLaunchKernelGGL(k_avgpool, dim3(1024), dim3(256), 0, st, x, o1, B, n.P, h, w, h2, w2);
        x = o1;
    }
    convnet_tail(st, n, B, x, dst_ptrs, dst_dense, pi, value);
}

// network.py:86-111: hidden rows src_ptrs[b] + action[b] -> normalised next hidden rows, reward [B], value [B], policy
// probabilities (optional: dead compute inside the reference search, mcts.py:386 expands with the root prior)
inline void convnet_recurrent(hipStream_t st, ConvNetDev& n, int B, const float* const* src_ptrs, const float* src_dense, const int* action,
                              float* const* dst_ptrs, float* dst_dense, float* reward, float* value, float* pi, const float* store_base = nullptr,
                              size_t store_floats = 0) {
    const int h = n.hh, w = n.hw, hw = h * w;
    if (n.dyn_inv_hw > 0) {  // board games: dense part over the real channels, then the <= 9 non-zero action-plane terms per output
        // the action terms ride in the conv's epilogue where a build for the geometry exists (bit-identical: the same additions in the same
        // order on the accumulator instead of on the stored value), else k_action_sparse adds them in place
        const ConvSparse sp{n.dyn_sp_w, n.dyn_sp_terms, action};
        const bool fused = conv_run(st, n.dyn_real, B, src_dense, src_ptrs, nullptr, 0, h, w, nullptr, n.bufA, true, store_base, store_floats,
                                    n.dyn_sp_terms ? &sp : nullptr);
        if (!fused) {
            ActionSparseLaunch S{};
            S.x = n.bufA; S.action = action; S.w = n.dyn_act_w; S.B = B; S.cout = n.P; S.h = h; S.w_img = w; S.A = n.A; S.inv_hw = n.dyn_inv_hw;
            hipLaunchKernelGGL(k_action_sparse, dim3((hw + 15) / 16, B), dim3(256), 0, st, S);
        }
    } else {
        conv_run(st, n.dyn_conv, B, src_dense, src_ptrs, action, n.A, h, w, nullptr, n.bufA, true, store_base, store_floats);
    }
    float* x = tower_run(st, n.dyn_res, 0, n.R, B, n.bufA, n.bufB, n.bufC, h, w, n.tw_dyn, n.tb_dyn);
    head_run(st, n.reward, B, x, nullptr, hw, 0, reward, nullptr);  // the reward head reads the un-normalised state (:447-448)
    convnet_tail(st, n, B, x, dst_ptrs, dst_dense, pi, value);
}

// ---------------------------------------------------------------------------------------------------------------------
// HBM-resident tree kernels: the tree_mode-0 device functions of mz_search.h on a per-block global region
// ---------------------------------------------------------------------------------------------------------------------
struct GTreeLaunch {
    SearchParams P;           // tree_mode 0 layout with the network part empty (offsets relative to the block region)
    unsigned char* regions;   // [blocks][P.lds_bytes]
    const float* pi0;         // [B][A]   (root policy from the initial inference)
    int hidden_size;
    const float** src_ptrs;   // [B] out of select: parent hidden rows
    float** dst_ptrs;         // [B] out of select: new node's hidden rows
    int* actions;             // [B] out of select
    const float* reward;      // in to backup: reward[env * rv_stride], value[env * rv_stride]
    const float* value;
    int rv_stride;
    int sim;
};

__global__ __launch_bounds__(256) void k_gtree_init(const GTreeLaunch G) {
    const SearchParams& P = G.P;
    unsigned char* smem = G.regions + (size_t)blockIdx.x * P.lds_bytes;
    const int tid = threadIdx.x, e = tid >> 4, a0 = tid & 15;
    const int env_g = blockIdx.x * TILE_E + e;
    const bool env_ok = env_g < P.B;
    double* ft = reinterpret_cast<double*>(smem + P.t_ftab);
    for (int i = tid; i < (P.S + 1) * (P.S + 1); i += 256) ft[i] = P.ftab[i];
    short* ch = reinterpret_cast<short*>(smem + P.t_child);
    for (int i = tid; i < TILE_E * P.NN * P.A; i += 256) ch[i] = -1;
    float* pi0 = reinterpret_cast<float*>(smem + P.t_pi0);
    for (int i = tid; i < TILE_E * P.A; i += 256) {
        const int ee = i / P.A, a = i - ee * P.A, eg = blockIdx.x * TILE_E + ee;
        pi0[i] = eg < P.B ? G.pi0[(size_t)eg * P.A + a] : 1.0f / (float)P.A;
    }
    if (a0 == 0) {
        TreeNode* r = node_at(smem, P, e, 0);
        r->W = 0.0; r->vq = 0.0; r->N = 0; r->reward = 0.0f; r->parent = -1; r->move = -1;
        r->player = env_ok ? P.cur[env_g] : 0;
        double* mm = reinterpret_cast<double*>(smem + P.t_mm) + e * 2;
        mm[0] = P.has_bounds ? P.kb_min : __longlong_as_double(0x7ff0000000000000LL);
        mm[1] = P.has_bounds ? P.kb_max : __longlong_as_double(0xfff0000000000000LL);
        int* sel = reinterpret_cast<int*>(smem + P.t_sel) + e * 4;
        sel[0] = sel[1] = sel[2] = sel[3] = 0;
    }
    root_noise_lanes(smem, P, e, a0, env_g, env_ok);
    __syncthreads();
    if (a0 == 0 && env_ok) root_prior(smem, P, e, env_g);
}

__global__ __launch_bounds__(256) void k_gtree_select(const GTreeLaunch G) {
    const SearchParams& P = G.P;
    unsigned char* smem = G.regions + (size_t)blockIdx.x * P.lds_bytes;
    const int tid = threadIdx.x, e = tid >> 4, a0 = tid & 15;
    const int env_g = blockIdx.x * TILE_E + e;
    const bool env_ok = env_g < P.B;
    int lp, la;
    tree_select<16>(smem, P, tid, env_ok, env_g, lp, la);
    if (a0 == 0 && env_ok) {
        float* base = P.hidden + (size_t)env_g * P.NN * G.hidden_size;
        G.src_ptrs[env_g] = base + (size_t)lp * G.hidden_size;
        G.dst_ptrs[env_g] = base + (size_t)(G.sim + 1) * G.hidden_size;
        G.actions[env_g] = la;
        if (P.trace_parent) {
            P.trace_parent[(size_t)env_g * P.S + G.sim] = lp;
            P.trace_action[(size_t)env_g * P.S + G.sim] = la;
        }
    }
}

// ---- select with ONE WAVE PER ENV (HBM trees, many actions) ----
// k_gtree_select gives an env 16 lanes and a workgroup 16 envs: at C5 (256 envs, 226 actions) that is 16 workgroups on 256 CUs, each
// lane walking 15 action chunks per level through dependent global loads -- 98 us per simulation, 2 % of a move.  Here a wave owns
// an env (64 lanes -> 4 chunks per level at 226 actions) and a 256-thread workgroup four envs: 4x fewer dependent chunks per level,
// 4x as many workgroups.  Same per-action arithmetic as select_level (child_Q + child_U, mcts.py:159-200), same tie set in
// ascending action order, same draw protocol: identical results.
__device__ __forceinline__ float wave_max_f32(float v) {
    v = butterfly16_max(v);
    const float a = __shfl_xor(v, 16, 64);
    v = a > v ? a : v;
    const float b = __shfl_xor(v, 32, 64);
    return b > v ? b : v;
}
__device__ __forceinline__ int nth_set_bit64(unsigned long long m, int idx) {
    for (int i = 0; i < idx; i++) m &= m - 1;
    return __ffsll((long long)m) - 1;
}
constexpr int MAX_CH64 = 4;  // action chunks of 64 lanes: A <= 256

__global__ __launch_bounds__(256) void k_gtree_select_wave(const GTreeLaunch G) {
    const SearchParams& P = G.P;
    const int lane = threadIdx.x & 63, env_g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (env_g >= P.B) return;  // (whole waves leave: no barrier in this kernel)
    unsigned char* smem = G.regions + (size_t)(env_g / TILE_E) * P.lds_bytes;
    const int e = env_g % TILE_E;
    double* mm = reinterpret_cast<double*>(smem + P.t_mm) + e * 2;
    int* sel = reinterpret_cast<int*>(smem + P.t_sel) + e * 4;
    const double* ftab = reinterpret_cast<const double*>(smem + P.t_ftab);
    const double* prior = reinterpret_cast<const double*>(smem + P.t_prior) + e * P.A;
    const double mn = mm[0], mx = mm[1];
    const bool norm = mx > mn, prior_f32 = (P.noise_mode == 0 && !P.legacy_promo);
    const int nch = (P.A + 63) >> 6;
    int n = 0, cp = P.cur[env_g], op = P.opp[env_g], ties = sel[3];
    int lp = 0, la = 0, lpl = 0, depth = 0;
    for (;;) {
        // ---- best_child of node n (mcts.py:104-127) ----
        const int Nn = node_at(smem, P, e, n)->N;
        const double* frow = ftab + Nn * (P.S + 1);
        const short* crow = child_row(smem, P, e, n);
        float u[MAX_CH64];
        float best = __uint_as_float(0xff800000u);
#pragma unroll
        for (int ch = 0; ch < MAX_CH64; ch++) {
            u[ch] = __uint_as_float(0xff800000u);
            const int a = ch * 64 + lane;
            if (ch < nch && a < P.A) {
                const int c = crow[a];
                int cn = 0;
                float qa = 0.0f;  // child_Q, mcts.py:159-178
                if (c >= 0) {
                    const TreeNode* cd = node_at(smem, P, e, c);
                    cn = cd->N;
                    if (cn > 0) {
                        double v = cd->vq;
                        if (norm) v = (v - mn) / (mx - mn);
                        qa = (float)v;
                    }
                }
                const double f = frow[cn];  // child_U, mcts.py:180-200
                const float ua = prior_f32 ? ((float)prior[a] * (float)f) : (float)(prior[a] * f);
                u[ch] = qa + ua;
                best = u[ch] > best ? u[ch] : best;
            }
        }
        best = wave_max_f32(best);
        unsigned long long msk[MAX_CH64];  // tie set in ascending action order (np.where(ucb == max), mcts.py:124)
        int total = 0;
#pragma unroll
        for (int ch = 0; ch < MAX_CH64; ch++) {
            const int a = ch * 64 + lane;
            msk[ch] = __ballot((ch < nch) && (a < P.A) && (u[ch] == best));
            total += __popcll(msk[ch]);
        }
        int pick = 0;
        if (total > 1) {  // np.random.choice consumes randomness only when there is a real tie (wave-uniform)
            double uu;
            if (P.rng_mode == 0) {
                if (ties < P.max_ties) uu = P.u_tie[(size_t)env_g * P.max_ties + ties];
                else { uu = 0.5; if (lane == 0) atomicExch(P.err, 4); }
            } else {
                Philox g(P.seed, P.env_offset + (unsigned)env_g, P.move_counter, 0x10000000u + (unsigned)ties);
                uu = g.uniform();
                if (P.dbg_utie && lane == 0 && ties < P.max_ties) P.dbg_utie[(size_t)env_g * P.max_ties + ties] = uu;
            }
            ties++;
            pick = (int)floor(uu * (double)total);
            pick = pick >= total ? total - 1 : pick;
        }
        int a_sel = 0, cum = 0;
        bool found = false;
#pragma unroll
        for (int ch = 0; ch < MAX_CH64; ch++) {
            const int c = __popcll(msk[ch]);
            if (!found && pick < cum + c) {
                a_sel = ch * 64 + nth_set_bit64(msk[ch], pick - cum);
                found = true;
            }
            cum += c;
        }
        const int t = cp; cp = op; op = t;  // mcts.py:379
        const int c = crow[a_sel];
        depth++;
        if (c < 0 || depth > P.NN) { lp = n; la = a_sel; lpl = cp; break; }
        n = c;
    }
    if (lane == 0) {
        sel[0] = lp; sel[1] = la; sel[2] = lpl; sel[3] = ties;
        float* base = P.hidden + (size_t)env_g * P.NN * G.hidden_size;
        G.src_ptrs[env_g] = base + (size_t)lp * G.hidden_size;
        G.dst_ptrs[env_g] = base + (size_t)(G.sim + 1) * G.hidden_size;
        G.actions[env_g] = la;
        if (P.trace_parent) {
            P.trace_parent[(size_t)env_g * P.S + G.sim] = lp;
            P.trace_action[(size_t)env_g * P.S + G.sim] = la;
        }
    }
}

__global__ __launch_bounds__(256) void k_gtree_backup(const GTreeLaunch G) {
    const SearchParams& P = G.P;
    unsigned char* smem = G.regions + (size_t)blockIdx.x * P.lds_bytes;
    const int tid = threadIdx.x, e = tid >> 4, a0 = tid & 15;
    const int env_g = blockIdx.x * TILE_E + e;
    if (a0 == 0 && env_g < P.B) tree_expand_backup(smem, P, e, G.sim, G.reward[(size_t)env_g * G.rv_stride], G.value[(size_t)env_g * G.rv_stride]);
}

__global__ __launch_bounds__(256) void k_gtree_finish(const GTreeLaunch G) {
    const SearchParams& P = G.P;
    unsigned char* smem = G.regions + (size_t)blockIdx.x * P.lds_bytes;
    const int tid = threadIdx.x, e = tid >> 4, a0 = tid & 15;
    const int env_g = blockIdx.x * TILE_E + e;
    if (a0 == 0 && env_g < P.B) tree_finish(smem, P, e, env_g);
}

}  // namespace mz
