// learner.hip -- host side of libmzlearner_hip.so: the C ABI of include/mzlearner.h over the gfx950 kernels of mz_learn.h.
// One handle == one GPU; every call enqueues kernels on the caller's stream and returns (no synchronisation, no allocation after create).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mzlearner.h"
#include "mz_learn.h"
#include "mz_learn_conv_host.h"

using namespace mzl;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
void mzl_internal_set_error(const std::string& msg) { g_err = msg; }  // (learner_replay.hip)
#define HIPCHK(expr)                                                                                                                     \
    do {                                                                                                                                 \
        hipError_t _e = (expr);                                                                                                          \
        if (_e != hipSuccess)                                                                                                            \
            return fail(MZL_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

static const char* kNames[NLAYER] = {
    "represent_net.net.0",        "represent_net.net.2",        "dynamics_net.transition_net.0", "dynamics_net.transition_net.2",
    "dynamics_net.reward_net.0",  "dynamics_net.reward_net.2",  "prediction_net.policy_net.0",   "prediction_net.policy_net.2",
    "prediction_net.value_net.0", "prediction_net.value_net.2",
};

struct mz_learner {
    mzl_config cfg{};
    int device = 0;
    mzlc_learner* conv = nullptr;  // net_kind == MZL_NET_BOARD: the conv-net learner (learner_conv.hip); every entry point forwards to it
    LNet net{};
    LSave sv{};
    LLds o{};
    LParams P{};
    std::string names[2 * NLAYER];
    int tiles_cap = 0;
    int lds_bytes = 0;
    LLds o_heads{};   // LDS layout / bytes of the heads-only launch (streaming form)
    int lds_heads = 0;
    std::vector<void*> allocs;
    DwJob* d_jobs = nullptr;
    DwBig* d_big = nullptr;     // one job per wave, up to 4 x 4 tiles: long reductions (k_learn_dw_big)
    std::vector<DwBig> big;
    std::vector<DwJob> jobs;   // host copy (R is patched per batch size)
    int jobs_tiles = -1;       // tile count the device copy of the jobs was built for
    float* d_sq = nullptr;     // partial sums of squares of the gradient
    int sq_blocks = 0;
    int num_cus = 256;
    const void* checked_ptr[9] = {};  // batch pointers already verified to be device memory (mzl_grad)
    int fast_max_tiles = 16, stage_fast_max_tiles = 256;  // MZL_FAST_MAX_TILES / MZL_CHAIN_FAST_MAX_TILES at create (see mzl_grad)
    int chain_min_tiles = 96, chain_max_tiles = 1 << 30;  // batch range of the persistent chain kernels (MZL_CHAIN_MIN_TILES / MZL_CHAIN_MAX_TILES at create)
    float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr;
    bool committed = false;
    int back_parts = 1;  // plane slices of the backward stages at small batches (k_learn_back_sliced), 1: unsliced
    bool fast = false;  // every GEMM of the net fits the register-resident forms: the kernels without their generic paths
};

static int tiles16(int x) { return (x + 15) / 16; }

extern "C" const char* mzl_last_error(void) { return g_err.c_str(); }

template <typename T>
static hipError_t dalloc(mz_learner* h, T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T) + 256);
    if (e != hipSuccess) return e;
    e = hipMemset(q, 0, count * sizeof(T) + 256);
    h->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return e;
}

static void build_jobs(mz_learner* h) {
    const LNet& n = h->net;
    const LSave& s = h->sv;
    struct Src { const float *a, *b; int a_ft, b_ft, steps; };
    const int xt = n.h_t + n.a_t;
    const Src src[NLAYER] = {
        {s.dz_rep0, s.in_rep, n.p_t, n.in_t, 1}, {s.dz_rep1, s.h1_rep, n.h_t, n.p_t, 1},
        {s.dz_dyn0, s.x, n.p_t, xt, n.K},        {s.dz_dyn1, s.h1_dyn, n.h_t, n.p_t, n.K},
        {s.dz_rew0, s.u_in, n.p_t, n.h_t, n.K},  {s.dz_rew1, s.h1_rew, n.sr_t, n.p_t, n.K},
        {s.dz_pol0, s.x, n.p_t, xt, n.K},        {s.dz_pol1, s.h1_pol, n.a_t, n.p_t, n.K},   // policy / value layer 1 read h_k: the first
        {s.dz_val0, s.x, n.p_t, xt, n.K},        {s.dz_val1, s.h1_val, n.sv_t, n.p_t, n.K},  // h_t tiles of the x blocks
    };
    h->jobs.clear();
    for (int l = 0; l < NLAYER; l++) {
        const LLayer& L = n.L[l];
        const int row_tiles = L.nt;
        const int col_tiles = (l == DYN0) ? xt : tiles16(L.k);
        DwJob j{};
        j.a = src[l].a; j.b = src[l].b; j.a_ft = src[l].a_ft; j.b_ft = src[l].b_ft;
        j.gw = h->P.t[2 * l].off; j.n = L.n; j.k = L.k;
        j.kH = h->P.kH[l]; j.kHpad = h->P.kHpad[l];
        j.R = src[l].steps;  // multiplied by the tile count at launch time
        if (row_tiles >= col_tiles) {  // one row tile x up to 4 column tiles
            for (int rt = 0; rt < row_tiles; rt++)
                for (int c0 = 0; c0 < col_tiles; c0 += 4) {
                    DwJob q = j;
                    q.a_t0 = rt; q.na = 1; q.b_t0 = c0; q.nb = col_tiles - c0 < 4 ? col_tiles - c0 : 4;
                    q.gb = c0 == 0 ? h->P.t[2 * l + 1].off : -1;
                    h->jobs.push_back(q);
                }
        } else {  // up to 4 row tiles x one column tile
            for (int ct = 0; ct < col_tiles; ct++)
                for (int r0 = 0; r0 < row_tiles; r0 += 4) {
                    DwJob q = j;
                    q.b_t0 = ct; q.nb = 1; q.a_t0 = r0; q.na = row_tiles - r0 < 4 ? row_tiles - r0 : 4;
                    q.gb = ct == 0 ? h->P.t[2 * l + 1].off : -1;
                    h->jobs.push_back(q);
                }
        }
    }
}

static void build_big_jobs(mz_learner* h) {
    h->big.clear();
    // the small jobs already enumerate (layer, operands): rebuild 4 x 5 tile groups from each layer's first job.  Order: layer, slice, unit,
    // each (layer, slice) padded to whole workgroups of DWB_WAVES units -- a workgroup's waves share one slice of one layer (mz_learn.h).
    // A layer of `steps` reduction blocks per tile gets grad_slices * steps / K slices: every workgroup then runs about the same number
    // of reduction blocks (the representation layers have 1 block per tile, the unrolled layers K).
    const LNet& n = h->net;
    for (int l = 0; l < NLAYER; l++) {
        const DwJob* first = nullptr;
        for (const auto& j : h->jobs)
            if (j.gw == h->P.t[2 * l].off) { first = &j; break; }
        const int row_tiles = n.L[l].nt, col_tiles = (l == DYN0) ? n.h_t + n.a_t : tiles16(n.L[l].k);
        std::vector<DwBig> units;
        for (int r0 = 0; r0 < row_tiles; r0 += 4)
            for (int c0 = 0; c0 < col_tiles;) {
                const int left = col_tiles - c0, nb = left <= DWB_NB ? left : 4;  // (a 5-tile row stays one unit; longer rows go in fours)
                DwBig q{};
                q.a = first->a; q.b = first->b; q.gw = first->gw; q.gb = c0 == 0 ? h->P.t[2 * l + 1].off : -1;
                q.a_ft = first->a_ft; q.b_ft = first->b_ft;
                q.a_t0 = r0; q.na = row_tiles - r0 < 4 ? row_tiles - r0 : 4;
                q.b_t0 = c0; q.nb = nb;
                q.R = first->R; q.n = first->n; q.k = first->k; q.kH = first->kH; q.kHpad = first->kHpad;
                units.push_back(q);
                c0 += nb;
            }
        int nj = (h->cfg.grad_slices * first->R + n.K - 1) / n.K;  // first->R: the layer's reduction blocks per tile (1 or K)
        nj = nj < 1 ? 1 : (nj > h->cfg.grad_slices ? h->cfg.grad_slices : nj);
        for (int k = 0; k < nj; k++) {
            for (auto q : units) {
                q.slice = k; q.nslices = nj;
                h->big.push_back(q);
            }
            while (h->big.size() % DWB_WAVES) h->big.push_back(DwBig{});  // (na == 0: the wave exits)
        }
    }
}

extern "C" int mzl_create(const mzl_config* cfg, int device_id, mz_learner** out) {
    if (!cfg || !out) return fail(MZL_E_INVALID, "null argument");
    if (cfg->net_kind != MZL_NET_MLP && cfg->net_kind != MZL_NET_BOARD && cfg->net_kind != MZL_NET_ATARI)
        return fail(MZL_E_INVALID, "net_kind must be MZL_NET_MLP, MZL_NET_BOARD or MZL_NET_ATARI");
    if (cfg->net_kind != MZL_NET_MLP) {
        if (cfg->in_dim < 1 || cfg->num_actions < 1 || cfg->num_actions > 32767 || cfg->num_planes < 1 || cfg->unroll_steps < 1 || cfg->unroll_steps > 32 ||
            cfg->max_batch < 1)
            return fail(MZL_E_INVALID, "bad learner dimensions");
        int ndev = 0;
        HIPCHK(hipGetDeviceCount(&ndev));
        if (ndev <= 0) return fail(MZL_E_HIP, "no HIP device visible: the learner kernels have no CPU fallback");
        if (device_id < 0 || device_id >= ndev) return fail(MZL_E_INVALID, "device_id out of range");
        HIPCHK(hipSetDevice(device_id));
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || cus <= 0) cus = 256;
        mzlc_learner* c = nullptr;
        std::string err;
        const int rc = mzlc_create(cfg, device_id, cus, &c, err);
        if (rc != MZL_OK) return fail(rc, err);
        mz_learner* h = new mz_learner();
        h->cfg = *cfg; h->device = device_id; h->conv = c;
        *out = h;
        return MZL_OK;
    }
    if (cfg->in_dim < 1 || cfg->num_actions < 1 || cfg->num_planes < 1 || cfg->hidden_dim < 1 || cfg->value_support_size < 1 ||
        cfg->reward_support_size < 1 || cfg->unroll_steps < 1 || cfg->unroll_steps > 32 || cfg->max_batch < 1 || cfg->grad_slices < 1 || cfg->grad_slices > 64)
        return fail(MZL_E_INVALID, "bad learner dimensions");
    if (cfg->num_actions > 32767) return fail(MZL_E_INVALID, "num_actions must fit int16");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(MZL_E_HIP, "no HIP device visible: the learner kernels have no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(MZL_E_INVALID, "device_id out of range");
    HIPCHK(hipSetDevice(device_id));
    mz_learner* h = new mz_learner();
    h->cfg = *cfg;
    h->device = device_id;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) h->num_cus = cus;
        if (getenv("MZL_FAST_MAX_TILES")) h->fast_max_tiles = atoi(getenv("MZL_FAST_MAX_TILES"));
        if (getenv("MZL_CHAIN_FAST_MAX_TILES")) h->stage_fast_max_tiles = atoi(getenv("MZL_CHAIN_FAST_MAX_TILES"));
        if (getenv("MZL_CHAIN_MIN_TILES")) h->chain_min_tiles = atoi(getenv("MZL_CHAIN_MIN_TILES"));
        if (getenv("MZL_CHAIN_MAX_TILES")) h->chain_max_tiles = atoi(getenv("MZL_CHAIN_MAX_TILES"));
    }
    LNet& n = h->net;
    n.in_dim = cfg->in_dim; n.A = cfg->num_actions; n.P = cfg->num_planes; n.H = cfg->hidden_dim;
    n.Sv = cfg->value_support_size; n.Sr = cfg->reward_support_size; n.K = cfg->unroll_steps;
    n.in_t = tiles16(n.in_dim); n.h_t = tiles16(n.H); n.a_t = tiles16(n.A); n.p_t = tiles16(n.P); n.sv_t = tiles16(n.Sv); n.sr_t = tiles16(n.Sr);
    int smax = n.a_t; smax = n.sv_t > smax ? n.sv_t : smax; smax = n.sr_t > smax ? n.sr_t : smax;
    n.lgs = smax * 16 + 1;
    const int dims[NLAYER][2] = {{n.P, n.in_dim}, {n.H, n.P}, {n.P, n.H + n.A}, {n.H, n.P}, {n.P, n.H},
                                 {n.Sr, n.P},     {n.P, n.H}, {n.A, n.P},       {n.P, n.H}, {n.Sv, n.P}};
    int off = 0;
    auto cleanup = [&](int rc) { (void)mzl_destroy(h); return rc; };
    for (int l = 0; l < NLAYER; l++) {
        LLayer& L = n.L[l];
        L.n = dims[l][0]; L.k = dims[l][1];
        L.nt = tiles16(L.n);
        L.kg = l == DYN0 ? n.h_t + n.a_t : tiles16(L.k);
        h->P.kH[l] = l == DYN0 ? n.H : L.k;
        h->P.kHpad[l] = l == DYN0 ? n.h_t * 16 : L.kg * 16;
        h->P.nt[l] = L.nt; h->P.kg[l] = L.kg;
        h->P.t[2 * l] = LTensor{off, L.n, L.k, l, 0};
        h->names[2 * l] = std::string(kNames[l]) + ".weight";
        off += L.n * L.k;
        h->P.t[2 * l + 1] = LTensor{off, L.n, 0, l, 1};
        h->names[2 * l + 1] = std::string(kNames[l]) + ".bias";
        off += L.n;
        const size_t wsz = (size_t)L.nt * L.kg * 256;
        if (dalloc(h, &h->P.wp[l], wsz) != hipSuccess || dalloc(h, &h->P.wtp[l], wsz) != hipSuccess || dalloc(h, &h->P.b[l], (size_t)L.nt * 16) != hipSuccess)
            return cleanup(fail(MZL_E_HIP, "hipMalloc failed (operand copies)"));
        L.wp = h->P.wp[l]; L.wtp = h->P.wtp[l]; L.b = h->P.b[l];
    }
    h->P.total = off;
    // LDS carve-out of the stage kernels
    LLds& o = h->o;
    int f = 0;
    const int xt = (n.in_t > n.h_t + n.a_t ? n.in_t : n.h_t + n.a_t);
    o.X = f; f += xt * 256;
    o.H1 = f; f += n.p_t * 256;
    o.DZ = o.H1;  // dL/dz1 takes the place of relu(z1): a head's forward layers are done with H1 before its backward pass writes DZ, the
                  // dynamics role never writes DZ and the backward sweep never reads H1 -- 32 KB less LDS at 512 planes: two workgroups per CU fit
    o.HN = f; f += n.h_t * 256;
    o.HS = f; f += n.h_t * 256;
    o.G = f; f += n.h_t * 256;
    o.R = f; f += n.h_t * 256;
    o.LG = f; f += (16 * n.lgs + 3) & ~3;
    o.DL = f; f += smax * 256;
    o.RED = f; f += LW * 256;
    o.MISC = f; f += 64;
    o.total = f;
    h->lds_bytes = f * 4;
    {   // the heads' launch of the streaming form: only the blocks a head role touches (three workgroups per CU fit at 512 planes)
        LLds& q = h->o_heads;
        q = o;
        int g = 0;
        q.X = g; g += n.h_t * 256;
        q.H1 = g; g += n.p_t * 256;
        q.DZ = q.H1;
        q.G = g; g += n.h_t * 256;
        q.LG = g; g += (16 * n.lgs + 3) & ~3;
        q.DL = g; g += smax * 256;
        q.RED = g; g += LW * 256;
        q.MISC = g; g += 64;
        q.HN = q.HS = q.R = 0;  // (never addressed by a head role)
        q.total = g;
        h->lds_heads = g * 4;
    }
    if (h->lds_bytes > 160 * 1024) return cleanup(fail(MZL_E_INVALID, "network needs " + std::to_string(h->lds_bytes) + " bytes of LDS per workgroup (> 160 KiB)"));
    // saved tensors and chain tensors
    const int tiles = tiles16(cfg->max_batch), K = n.K;
    h->tiles_cap = tiles;
    LSave& s = h->sv;
    const size_t T = (size_t)tiles * 256;
    bool ok = true;
    auto A = [&](float** p, size_t count) { ok = ok && dalloc(h, p, count) == hipSuccess; };
    A(&s.in_rep, T * n.in_t); A(&s.h1_rep, T * n.p_t); A(&s.dz_rep0, T * n.p_t); A(&s.dz_rep1, T * n.h_t);
    A(&s.x, T * K * (n.h_t + n.a_t)); A(&s.h1_dyn, T * K * n.p_t); A(&s.dz_dyn0, T * K * n.p_t); A(&s.dz_dyn1, T * K * n.h_t);
    A(&s.u_in, T * K * n.h_t); A(&s.h1_rew, T * K * n.p_t); A(&s.dz_rew0, T * K * n.p_t); A(&s.dz_rew1, T * K * n.sr_t);
    A(&s.h1_pol, T * K * n.p_t); A(&s.dz_pol0, T * K * n.p_t); A(&s.dz_pol1, T * K * n.a_t);
    A(&s.h1_val, T * K * n.p_t); A(&s.dz_val0, T * K * n.p_t); A(&s.dz_val1, T * K * n.sv_t);
    A(&s.hc, T * (K + 1) * n.h_t); A(&s.uc, T * (K + 1) * n.h_t);
    A(&s.up, T * (K + 1) * n.h_t * DX_PARTS);
    A(&s.dxd, T * K * n.h_t * DX_PARTS); A(&s.dxp, T * K * n.h_t); A(&s.dxv, T * K * n.h_t); A(&s.dxr, T * K * n.h_t);
    A(&s.lossp, (size_t)3 * K * tiles);
    if (getenv("MZL_STAMPS")) ok = ok && dalloc(h, &s.stamps, 64) == hipSuccess;
    ok = ok && dalloc(h, &s.actc, (size_t)K * tiles * 16) == hipSuccess;
    h->sq_blocks = (h->P.total + 255) / 256;
    A(&h->d_sq, (size_t)h->sq_blocks);
    if (!ok) return cleanup(fail(MZL_E_HIP, "hipMalloc failed (saved tensors)"));
    {   // the shapes the <F = true> kernels assume (mz_learn.h: wide_load / ks_load call sites, chain_ld)
        auto ks_ok = [&](int nt, int kg) { return 2 * nt <= LW && (kg + LW / nt - 1) / (LW / nt) <= KSB; };
        h->fast = n.p_t <= 4 * LW && n.in_t <= 6 && n.h_t + n.a_t <= 6 && n.h_t <= 4 && n.a_t <= 2 && n.sv_t <= 2 && n.sr_t <= 2 && n.h_t * 64 <= LT &&
                  ks_ok(n.h_t, n.p_t) && ks_ok(n.a_t, n.p_t) && ks_ok(n.sv_t, n.p_t) && ks_ok(n.sr_t, n.p_t) && !getenv("MZL_GENERIC");
    }
    {   // k_learn_back_sliced: four slices when the register forms fit them (p_t / 4 plane tiles per workgroup, one per wave)
        const bool ok = h->fast && n.p_t % DX_PARTS == 0 && n.p_t / DX_PARTS <= LW && n.h_t <= 4 && LW % n.h_t == 0 && (n.p_t / DX_PARTS) % (LW / n.h_t) == 0 &&
                        (n.p_t / DX_PARTS) / (LW / n.h_t) <= 4 && n.h_t * 64 <= LT && !getenv("MZL_NO_SLICE");
        h->back_parts = ok ? DX_PARTS : 1;
    }
    build_jobs(h);
    build_big_jobs(h);
    if (dalloc(h, &h->d_jobs, h->jobs.size()) != hipSuccess || dalloc(h, &h->d_big, h->big.size()) != hipSuccess) return cleanup(fail(MZL_E_HIP, "hipMalloc failed (jobs)"));
    hipError_t e = hipSuccess;
    const void* stage_kernels[] = {(const void*)&k_learn_repr<true>,   (const void*)&k_learn_repr<false>, (const void*)&k_learn_unroll<true>,
                                   (const void*)&k_learn_unroll<false>, (const void*)&k_learn_back<true>,  (const void*)&k_learn_back<false>,
                                   (const void*)&k_learn_back_sliced<true>, (const void*)&k_learn_fwd_sliced<true>,
                                   (const void*)&k_learn_unroll<false, 3>, (const void*)&k_learn_dyn_chain, (const void*)&k_learn_dyn_back_chain};
    // the attribute belongs to the kernel symbol, not to this handle: a later, smaller learner in the same process must not lower the limit of a
    // live larger one -- every handle sets the hardware maximum
    for (const void* f : stage_kernels)
        if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_learn_dw), hipFuncAttributeMaxDynamicSharedMemorySize, LW * 8 * 256 * 4);
    if (e != hipSuccess) return cleanup(fail(MZL_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)));
    e = hipDeviceSynchronize();  // (dalloc's fills run on the NULL stream; the caller's stream need not wait for it)
    if (e != hipSuccess) return cleanup(fail(MZL_E_HIP, std::string("hipDeviceSynchronize: ") + hipGetErrorString(e)));
    *out = h;
    return MZL_OK;
}

extern "C" int mzl_destroy(mz_learner* h) {
    if (!h) return MZL_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (h->conv) mzlc_destroy(h->conv);
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
    return MZL_OK;
}

// diagnostic: device pointer of the cycle stamps (MZL_STAMPS=1 at create), or NULL
extern "C" void* mzl_debug_stamps(const mz_learner* h) { return h ? h->sv.stamps : nullptr; }

extern "C" int64_t mzl_num_params(const mz_learner* h) { return !h ? 0 : (h->conv ? mzlc_num_params(h->conv) : h->P.total); }
extern "C" int64_t mzl_grad_floats(const mz_learner* h) {
    return !h ? 0 : (h->conv ? mzlc_num_params(h->conv) : (int64_t)h->P.total * h->cfg.grad_slices);
}
extern "C" int32_t mzl_num_tensors(const mz_learner* h) { return !h ? 0 : (h->conv ? mzlc_num_tensors(h->conv) : 2 * NLAYER); }
extern "C" int32_t mzl_num_buffers(const mz_learner* h) { return (h && h->conv) ? mzlc_num_buffers(h->conv) : 0; }
extern "C" int64_t mzl_num_running(const mz_learner* h) { return (h && h->conv) ? mzlc_num_running(h->conv) : 0; }
extern "C" int mzl_buffer_info(const mz_learner* h, int32_t i, const char** name, int64_t* offset, int32_t* count) {
    if (!h || !h->conv || mzlc_buffer_info(h->conv, i, name, offset, count) != MZL_OK) return fail(MZL_E_INVALID, "buffer index out of range");
    return MZL_OK;
}
extern "C" int mzl_bind_buffers(mz_learner* h, float* d_running, int64_t* d_num_batches) {
    if (!h) return fail(MZL_E_INVALID, "null learner");
    if (!h->conv) return MZL_OK;  // (MLP nets have no buffers)
    if (!d_running || !d_num_batches) return fail(MZL_E_INVALID, "null argument to mzl_bind_buffers");
    return mzlc_bind_buffers(h->conv, d_running, d_num_batches);
}
// diagnostic (tests): device pointers of the conv learner's saved tensors (learner_conv.hip mzlc_debug_tensor)
extern "C" int mzl_debug_tensor(const mz_learner* h, const char* what, int a, int b, void** ptr, int64_t* count) {
    if (!h || !h->conv || !what || !ptr || !count || mzlc_debug_tensor(h->conv, what, a, b, ptr, count) != MZL_OK) return fail(MZL_E_INVALID, "no such tensor");
    return MZL_OK;
}

extern "C" int mzl_tensor_info(const mz_learner* h, int32_t i, const char** name, int64_t* offset, int32_t* rows, int32_t* cols) {
    if (h && h->conv) {
        if (mzlc_tensor_info(h->conv, i, name, offset, rows, cols) != MZL_OK) return fail(MZL_E_INVALID, "tensor index out of range");
        return MZL_OK;
    }
    if (!h || i < 0 || i >= 2 * NLAYER) return fail(MZL_E_INVALID, "tensor index out of range");
    const LTensor& t = h->P.t[i];
    if (name) *name = h->names[i].c_str();
    if (offset) *offset = t.off;
    if (rows) *rows = t.n;
    if (cols) *cols = t.k;
    return MZL_OK;
}

extern "C" int mzl_bind(mz_learner* h, float* d_params, float* d_grads, float* d_exp_avg, float* d_exp_avg_sq) {
    if (!h || !d_params || !d_grads || !d_exp_avg || !d_exp_avg_sq) return fail(MZL_E_INVALID, "null argument to mzl_bind");
    if (h->conv) return mzlc_bind(h->conv, d_params, d_grads, d_exp_avg, d_exp_avg_sq);
    h->params = d_params; h->grads = d_grads; h->m = d_exp_avg; h->v = d_exp_avg_sq;
    h->committed = false;
    h->jobs_tiles = -1;  // (the job tables are re-uploaded and the new gradient slices zeroed by the next mzl_grad)
    return MZL_OK;
}

static int launch_adam(mz_learner* h, const AdamArgs& a, hipStream_t st) {
    int maxcnt = 0;
    for (int i = 0; i < 2 * NLAYER; i++) {
        const LTensor& t = h->P.t[i];
        const int c = t.is_bias ? t.n : t.n * t.k;
        maxcnt = c > maxcnt ? c : maxcnt;
    }
    hipLaunchKernelGGL(k_learn_adam, dim3((maxcnt + 255) / 256, 2 * NLAYER), dim3(256), 0, st, h->P, h->params, h->grads, h->m, h->v, h->d_sq, a);
    HIPCHK(hipGetLastError());
    return MZL_OK;
}

extern "C" int mzl_commit(mz_learner* h, void* stream) {
    if (!h) return fail(MZL_E_INVALID, "null learner");
    if (h->conv) {
        std::string err;
        const int rc = mzlc_commit(h->conv, stream, err);
        return rc == MZL_OK ? MZL_OK : fail(rc, err);
    }
    if (!h->params) return fail(MZL_E_STATE, "mzl_bind first");
    HIPCHK(hipSetDevice(h->device));
    AdamArgs a{};
    a.pack_only = 1;
    const int rc = launch_adam(h, a, reinterpret_cast<hipStream_t>(stream));
    if (rc) return rc;
    h->committed = true;
    return MZL_OK;
}

extern "C" int mzl_grad(mz_learner* h, const mzl_batch* b, void* stream) {
    if (!h || !b) return fail(MZL_E_INVALID, "null argument to mzl_grad");
    if (h->conv) {
        std::string err;
        const int rc = mzlc_grad(h->conv, b, stream, err);
        return rc == MZL_OK ? MZL_OK : fail(rc, err);
    }
    if (!h->committed) return fail(MZL_E_STATE, "weights not committed: mzl_bind, then mzl_commit");
    if (b->batch < 1 || b->batch > h->cfg.max_batch) return fail(MZL_E_INVALID, "batch must be in [1, max_batch]");
    if (!b->d_index) return fail(MZL_E_INVALID, "d_index is required (rows 0 .. batch-1 for a stacked batch)");
    if (!b->d_state || !b->d_action || !b->d_pi_prob || !b->d_value || !b->d_reward || !b->d_weights || !b->d_loss || !b->d_priorities)
        return fail(MZL_E_INVALID, "null batch pointer");
    if (b->action_bytes != 1 && b->action_bytes != 2) return fail(MZL_E_INVALID, "action_bytes must be 1 or 2");
    {   // every pointer must be memory of this GPU (a host pointer would fault inside a kernel: checked here, once per distinct value)
        const void* ptrs[9] = {b->d_state, b->d_action, b->d_pi_prob, b->d_value, b->d_reward, b->d_index, b->d_weights, b->d_loss, b->d_priorities};
        static const char* names[9] = {"d_state", "d_action", "d_pi_prob", "d_value", "d_reward", "d_index", "d_weights", "d_loss", "d_priorities"};
        for (int i = 0; i < 9; i++) {
            if (ptrs[i] == h->checked_ptr[i]) continue;
            hipPointerAttribute_t at{};
            const hipError_t e = hipPointerGetAttributes(&at, ptrs[i]);
            if (e != hipSuccess || at.type != hipMemoryTypeDevice || at.device != h->device) {
                (void)hipGetLastError();
                return fail(MZL_E_INVALID, std::string(names[i]) + " is not memory of the learner's GPU (the batch is read where the replay lives: keep it in HBM)");
            }
            h->checked_ptr[i] = ptrs[i];
        }
    }
    if (b->action_bytes == 1 && h->cfg.num_actions > 128) return fail(MZL_E_INVALID, "num_actions > 128 needs int16 actions");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const LNet& n = h->net;
    LBatch bt{};
    bt.state = b->d_state; bt.action = b->d_action; bt.pi = b->d_pi_prob; bt.value = b->d_value; bt.reward = b->d_reward;
    bt.idx = b->d_index; bt.w = b->d_weights; bt.prio = b->d_priorities; bt.B = b->batch; bt.tiles = tiles16(b->batch);
    bt.state_i8 = b->state_is_int8; bt.action_bytes = b->action_bytes;
    const int tiles = bt.tiles, K = n.K;
    if (h->jobs_tiles != tiles) {  // the reduction length of the weight-gradient jobs follows the batch size
        std::vector<DwJob> jj = h->jobs;
        for (auto& j : jj) j.R *= tiles;
        std::vector<DwBig> bb = h->big;
        for (auto& j : bb) j.R *= tiles;
        HIPCHK(hipMemcpyAsync(h->d_jobs, jj.data(), jj.size() * sizeof(DwJob), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(h->d_big, bb.data(), bb.size() * sizeof(DwBig), hipMemcpyHostToDevice, st));
        // (a job with fewer units than slices never writes its part of the upper slices: those stay the zeros written here)
        HIPCHK(hipMemsetAsync(h->grads, 0, (size_t)h->P.total * h->cfg.grad_slices * sizeof(float), st));
        HIPCHK(hipStreamSynchronize(st));  // (temporaries; happens once per batch size)
        h->jobs_tiles = tiles;
    }
    const size_t lds = (size_t)h->lds_bytes;
    // register-resident operands (230 VGPRs: one workgroup per CU) pay for latency-bound grids of at most one workgroup per CU; deeper
    // grids run the streaming builds (60-94 VGPRs: two workgroups per CU, the heads-only build three -- one's loss rows and barriers under
    // the others' MFMAs).  The heads' grid is 15 x tiles: with the three-per-CU build the streaming form wins from 16 tiles on
    // (batch 512: 0.153 -> 0.146 ms, 1024: 0.202 -> 0.188, 2000: 0.317 -> 0.287; with two per CU it only won from ~2 k samples)
    const int fast_max_tiles = h->fast_max_tiles;
    // The CHAIN stages (representation, dynamics_k, their backward stages) are grid (tiles): up to one workgroup per CU they are latency
    // chains whatever the build, and the register-resident form is the shorter chain; the heads' grid is 15 x tiles deep, so they switch
    // to the streaming builds (two or three workgroups per CU) much earlier.
    const int chain_max_tiles = h->stage_fast_max_tiles;
    const bool fast_heads = h->fast && tiles < fast_max_tiles;
    const bool fast_chain_base = h->fast && (fast_heads || tiles <= chain_max_tiles);
    // the persistent chain kernels (register-resident, one workgroup per CU looping over its tiles) from `chain_min_tiles` on; below that the
    // plane-sliced stage kernels spread a stage over more CUs than there are tiles
    const bool chain = h->fast && n.h_t * 64 <= LT && tiles >= h->chain_min_tiles && tiles <= h->chain_max_tiles;
    const int sliced_ok = (h->back_parts > 1 && tiles * h->back_parts <= 512) ? h->back_parts : 1;
    const bool fast_chain = fast_chain_base || chain;
    const int fparts = (!chain && fast_chain && sliced_ok > 1 && n.in_t <= WKG && n.h_t + n.a_t <= WKG) ? h->back_parts : 1;
    // forward chain
    if (fast_chain) {
        if (fparts > 1) {  // small batches: the forward chain cut across the planes too (k = -1: representation; k = K: finishes u_K)
            for (int k = -1; k < K; k++) hipLaunchKernelGGL(k_learn_fwd_sliced<true>, dim3(tiles, fparts), dim3(LT), lds, st, n, h->sv, bt, h->o, k, fparts);
            hipLaunchKernelGGL(k_learn_fwd_sliced<true>, dim3(tiles, 1), dim3(LT), lds, st, n, h->sv, bt, h->o, K, fparts);
        } else if (chain) {  // one persistent workgroup per CU runs the whole dynamics chain of its tiles, operands loaded once (mz_learn.h)
            // (the representation stage is a plain grid over the tiles: register form up to one workgroup per CU, streaming form beyond)
            if (fast_chain_base) hipLaunchKernelGGL(k_learn_repr<true>, dim3(tiles), dim3(LT), lds, st, n, h->sv, bt, h->o);
            else hipLaunchKernelGGL(k_learn_repr<false>, dim3(tiles), dim3(LT), lds, st, n, h->sv, bt, h->o);
            hipLaunchKernelGGL(k_learn_dyn_chain, dim3(tiles < h->num_cus ? tiles : h->num_cus), dim3(LT), lds, st, n, h->sv, bt, h->o);
        } else {
            hipLaunchKernelGGL(k_learn_repr<true>, dim3(tiles), dim3(LT), lds, st, n, h->sv, bt, h->o);
            for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_learn_unroll<true>, dim3(tiles), dim3(LT), lds, st, n, h->sv, bt, h->o, k, 0);
        }
    } else {
        hipLaunchKernelGGL(k_learn_repr<false>, dim3(tiles), dim3(LT), lds, st, n, h->sv, bt, h->o);
        for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_learn_unroll<false>, dim3(tiles), dim3(LT), lds, st, n, h->sv, bt, h->o, k, 0);
    }
    // heads: forward + loss + backward of the three heads of every step
    if (fast_heads) {
        hipLaunchKernelGGL(k_learn_unroll<true>, dim3(tiles, 3, K), dim3(LT), lds, st, n, h->sv, bt, h->o, 0, 1);
    } else {
        static const bool heads3_on = !getenv("MZL_NO_HEADS3");
        const bool heads3 = heads3_on && n.p_t <= 4 * LW && n.a_t <= 2 && n.sv_t <= 2 && n.sr_t <= 2 && 3 * (h->lds_heads + 512) <= 160 * 1024;
        if (heads3) hipLaunchKernelGGL((k_learn_unroll<false, 3>), dim3(tiles, 3, K), dim3(LT), (size_t)h->lds_heads, st, n, h->sv, bt, h->o_heads, 0, 1);
        else hipLaunchKernelGGL(k_learn_unroll<false>, dim3(tiles, 3, K), dim3(LT), lds, st, n, h->sv, bt, h->o, 0, 1);
    }
    // backward chain
    if (fast_chain) {
        // small batches: the backward chain cut four ways across the planes (32 instead of 8 workgroups per stage at batch 128)
        LSave svb = h->sv;
        svb.dx_parts = chain ? 1 : sliced_ok;
        if (svb.dx_parts > 1) {
            for (int k = K - 1; k >= -1; k--) hipLaunchKernelGGL(k_learn_back_sliced<true>, dim3(tiles, svb.dx_parts), dim3(LT), lds, st, n, svb, bt, h->o, k);
        } else if (chain) {
            hipLaunchKernelGGL(k_learn_dyn_back_chain, dim3(tiles < h->num_cus ? tiles : h->num_cus), dim3(LT), lds, st, n, svb, bt, h->o);
            if (fast_chain_base) hipLaunchKernelGGL(k_learn_back<true>, dim3(tiles), dim3(LT), lds, st, n, svb, bt, h->o, -1);
            else hipLaunchKernelGGL(k_learn_back<false>, dim3(tiles), dim3(LT), lds, st, n, svb, bt, h->o, -1);
        } else {
            for (int k = K - 1; k >= -1; k--) hipLaunchKernelGGL(k_learn_back<true>, dim3(tiles), dim3(LT), lds, st, n, svb, bt, h->o, k);
        }
    } else {
        for (int k = K - 1; k >= -1; k--) hipLaunchKernelGGL(k_learn_back<false>, dim3(tiles), dim3(LT), lds, st, n, h->sv, bt, h->o, k);
    }
    // long reductions (grad_slices > 1 is the caller's statement that the batch is large): 4 x 4 tiles per wave, the slices carry the
    // parallelism; short ones: 1 x 4 tiles per workgroup, its eight waves split the reduction
    const int n_loss = 3 * K * tiles;
    if (h->cfg.grad_slices > 1 && tiles * K >= 64)
        hipLaunchKernelGGL(k_learn_dw_big, dim3(((unsigned)h->big.size() + DWB_WAVES - 1) / DWB_WAVES + 1), dim3(DWB_WAVES * 64), 0, st,
                           h->d_big, (int)h->big.size(), h->grads, (size_t)h->P.total, h->sv.lossp, n_loss, b->batch, b->d_loss);
    else
        hipLaunchKernelGGL(k_learn_dw, dim3((unsigned)h->jobs.size() + 1, h->cfg.grad_slices), dim3(LT), LW * 8 * 256 * 4, st, h->d_jobs, h->grads,
                           (size_t)h->P.total, h->sv.lossp, n_loss, b->batch, b->d_loss);
    if (h->cfg.grad_slices > 1)  // slice 0 <- the complete gradient (what a data-parallel learner all-reduces)
        hipLaunchKernelGGL(k_learn_gradsum, dim3(h->sq_blocks), dim3(256), 0, st, h->grads, (size_t)h->P.total, h->cfg.grad_slices, h->P.total, h->d_sq);
    HIPCHK(hipGetLastError());
    return MZL_OK;
}

extern "C" int mzl_apply(mz_learner* h, double lr, double beta1, double beta2, double eps, double weight_decay, double max_grad_norm, int64_t step,
                         void* stream) {
    if (!h) return fail(MZL_E_INVALID, "null learner");
    if (step < 1) return fail(MZL_E_INVALID, "Adam step numbers start at 1");
    if (h->conv) {
        std::string err;
        const int rc = mzlc_apply(h->conv, lr, beta1, beta2, eps, weight_decay, max_grad_norm, step, stream, err);
        return rc == MZL_OK ? MZL_OK : fail(rc, err);
    }
    if (!h->committed) return fail(MZL_E_STATE, "weights not committed");
    if (step < 1) return fail(MZL_E_INVALID, "Adam step numbers start at 1");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool clip = max_grad_norm > 0.0;
    if (clip) {  // partial sums of squares of the (possibly all-reduced) gradient in slice 0
        hipLaunchKernelGGL(k_learn_gradsum, dim3(h->sq_blocks), dim3(256), 0, st, h->grads, (size_t)h->P.total, 1, h->P.total, h->d_sq);
        HIPCHK(hipGetLastError());
    }
    AdamArgs a{};
    a.lr = (float)lr; a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps; a.weight_decay = (float)weight_decay;
    a.max_norm = clip ? (float)max_grad_norm : 0.0f;
    a.bc1 = (float)(1.0 - std::pow(beta1, (double)step));
    a.bc2 = (float)(1.0 - std::pow(beta2, (double)step));
    a.sq_blocks = h->sq_blocks;
    a.pack_only = 0;
    return launch_adam(h, a, st);
}
