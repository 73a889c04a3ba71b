// planner.hip -- host side of libmzplanner_hip.so: the C ABI of include/mzplanner.h over the gfx950 kernels
// in mz_search.h / mz_mlp.h / mz_env.h (MLP nets: one fused LDS-resident kernel per move) and mz_conv.h / mz_convnet.h
// (conv nets: HBM-resident trees + MFMA conv towers, a short kernel sequence per simulation).
// One planner handle == one GPU, one HIP stream, all state in HBM.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/mzplanner.h"
#include "mz_env.h"
#include "mz_search.h"
#include "mz_search_fast.h"
#include "mz_convnet.h"

using namespace mz;

static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess)                                                                                 \
            return fail(MZ_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                                      std::to_string(__LINE__) + ")");                                       \
    } while (0)

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

static const char* kMlpNames[L_COUNT] = {
    "represent_net.net.0",          "represent_net.net.2",          "dynamics_net.transition_net.0", "dynamics_net.transition_net.2",
    "dynamics_net.reward_net.0",    "dynamics_net.reward_net.2",    "prediction_net.policy_net.0",   "prediction_net.policy_net.2",
    "prediction_net.value_net.0",   "prediction_net.value_net.2",
};

struct mz_planner {
    mz_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    std::map<std::string, HostTensor> params;
    bool committed = false;

    MlpNet net{};
    MlpLds o{};
    float* d_w[L_COUNT] = {};
    float* d_b[L_COUNT] = {};
    SearchParams sp{};   // tree_mode 0 layout
    SearchParams sp2{};  // tree_mode 2 layout (valid if tree2_ok)
    bool tree2_ok = false;
    int lds_mode0 = 0, lds_mode2 = 0;
    // tree_mode 2 behind the TUNED kernel's own, smaller network carve-out (k_search_fast keeps the num_planes-wide layers in registers: of
    // the generic H1 / V1 buffers it uses 16 KiB each, for its K-split partial tiles): lets four-action, 50-simulation searches at
    // num_planes 512 (the LunarLander-shaped configuration) fit 160 KiB.  Only used when the generic carve-out does not fit.
    SearchParams sp2f{};
    bool tree2f_ok = false;
    int lds_mode2f = 0, fast_delta = 0;
    double* d_ftab_tri = nullptr;
    InferParams ip{};
    // tuned kernel for the benchmark shapes (mz_search_fast.h): per-wave weight streams of the wide layers
    int fast_planes = 0;  // 0: generic kernel only; 256 / 512: k_search_fast<P>
    bool tree_old = false;       // MZ_TREE_OLD=1: evaluate every level on every descent (A/B measurements, tests)
    bool force_generic = false;  // MZ_FORCE_GENERIC=1: run the shape-generic kernel (A/B measurements, tests)
    bool fuse_env = false;       // device self-play as one kernel per move instead of three (MZ_FUSE_ENV=0/1 overrides the default)
    int hwx = -1;  // k_search_fast helper-wave work split (MZ_HWX=0..3 overrides the default: A/B measurements)
    bool gtree_wave = true;   // HBM trees: select with one wave per env
    bool no_fast_layout = false;  // MZ_NO_FAST_LAYOUT=1: never give k_search_fast its own LDS carve-out (diagnostic)
    bool fast_ac4 = true;         // MZ_FAST_AC4=0: the general build instead of the four-action one (diagnostic)
    std::string last_dispatch = "none yet";  // what the last search launch ran (mz_planner_describe)
    float* d_stream[1] = {};
    float* d_bias_all = nullptr;
    double *d_dbg_noise = nullptr, *d_dbg_utie = nullptr, *d_dbg_ufinal = nullptr;  // mz_debug_capture_rng
    FastWeights fw{};

    // per-env device buffers (capacity cfg.num_envs)
    float* d_obs = nullptr;
    unsigned char* d_mask = nullptr;
    int *d_cur = nullptr, *d_opp = nullptr;
    double* d_temp = nullptr;
    double *d_noise = nullptr, *d_utie = nullptr, *d_ufinal = nullptr;
    float* d_hidden = nullptr;
    double* d_ftab = nullptr;
    int* d_action = nullptr;
    double *d_pi = nullptr, *d_root = nullptr;
    int* d_visits = nullptr;
    int* d_err = nullptr;
    long long* d_stamps = nullptr;
    // scripted hook
    float *d_spi0 = nullptr, *d_svalues = nullptr, *d_srewards = nullptr;
    int *d_tparent = nullptr, *d_taction = nullptr;
    // inference API buffers
    float *d_inf_in = nullptr, *d_inf_hidden = nullptr, *d_inf_reward = nullptr, *d_inf_value = nullptr, *d_inf_pi = nullptr;
    int* d_inf_action = nullptr;
    int inf_cap = 0;

    // conv nets (MZ_NET_BOARD / MZ_NET_ATARI): network, HBM tree regions and the per-simulation exchange buffers
    bool conv = false;
    // MLP nets whose trees do not fit the LDS-resident kernels (many simulations, > 64 actions; MZ_HBM_TREE=1 forces it): the
    // HBM tree kernels around k_infer launches, one (select, inference, backup) triple per simulation
    bool hbm_tree = false;
    SearchParams spg{};  // tree layout inside a per-workgroup HBM region (hbm_tree)
    float* d_pi_scratch = nullptr;
    ConvNetDev cnet{};
    unsigned char* d_regions = nullptr;
    float *d_pi0 = nullptr, *d_sim_reward = nullptr, *d_sim_value = nullptr;
    const float** d_srcptrs = nullptr;
    float **d_dstptrs = nullptr, **d_rootptrs = nullptr;
    int* d_sim_action = nullptr;

    // self-play
    EnvState env{};
    int env_kind = MZ_ENV_NONE;
    unsigned int move_counter = 0;
    int ring_len = 0, ring_pos = 0, ring_count = 0;
    bool has_replay = false;
    ReplayRing replay{};
    int* d_epi_off = nullptr;        // per-env slot offsets of the move being written (k_epi_scan)
    long long* d_epi_ctr = nullptr;  // reserved write cursor of the attached replay ring (k_epilogue reserves, k_epi_publish commits)
    long long selfplay_moves = 0;  // moves since mz_selfplay_reset

    // profiling
    bool profiling = false;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> kev;
    size_t kev_used = 0;
};

static int obs_dim(const mz_config& c) { return c.obs_c * c.obs_h * c.obs_w; }
static int pad16(int x) { return (x + 15) & ~15; }

extern "C" const char* mz_last_error(void) { return g_err.c_str(); }
extern "C" const char* mz_version(void) { return "mzplanner 0.1 (gfx950)"; }

// Which kernel build this handle's last search launch dispatched to, and every diagnostic switch as this handle read it (mzplanner.h)
static thread_local std::string g_describe;
extern "C" const char* mz_planner_describe(mz_planner* p) {
    if (!p) return "null planner";
    char b[512];
    const ConvSwitches& cs = conv_switches();
    snprintf(b, sizeof b,
             "; switches: MZ_FORCE_GENERIC=%d MZ_FUSE_ENV=%d MZ_GTREE_WAVE=%d MZ_HWX=%d MZ_TREE_OLD=%d MZ_HBM_TREE=%d MZ_NO_FAST_LAYOUT=%d MZ_FAST_AC4=%d"
             " | per process: MZ_ACTION_SPARSE=%d MZ_ACTION_FUSE=%d MZ_CONV_SPEC=%d MZ_TOWER=%d MZ_CONV_TILE=%d MZ_CONV_G=%d MZ_CONV_NCT=%d",
             (int)p->force_generic, (int)p->fuse_env, (int)p->gtree_wave, p->hwx, (int)p->tree_old, (int)p->hbm_tree, (int)p->no_fast_layout, (int)p->fast_ac4,
             cs.action_sparse, cs.action_fuse, cs.conv_spec, cs.tower, cs.conv_tile, cs.conv_g, cs.conv_nct);
    g_describe = "search: " + p->last_dispatch + b;
    return g_describe.c_str();
}

static void compute_layout(mz_planner* p) {
    const mz_config& c = p->cfg;
    MlpNet& n = p->net;
    n.in_dim = obs_dim(c); n.A = c.num_actions; n.P = c.num_planes; n.H = c.hidden_dim;
    n.Sv = c.value_support_size; n.Sr = c.reward_support_size;
    n.in_pad = pad16(n.in_dim); n.h_pad = pad16(n.H); n.x_pad = n.h_pad + pad16(n.A); n.p_pad = pad16(n.P);
    const int dims[L_COUNT][2] = {{n.P, n.in_dim}, {n.H, n.P}, {n.P, n.H + n.A}, {n.H, n.P}, {n.P, n.H},
                                  {n.Sr, n.P},     {n.P, n.H}, {n.A, n.P},       {n.P, n.H}, {n.Sv, n.P}};
    for (int l = 0; l < L_COUNT; l++) {
        MlpLayer& L = n.L[l];
        L.n = dims[l][0]; L.k = dims[l][1];
        L.n_tiles = (L.n + 15) / 16;
        if (l == L_DYN0) {  // [hidden padded to 16 | one-hot action block(s)], summation order: mz_mlp.h header
            const int ab = (n.A + 15) / 16, a_last = n.A - 16 * (ab - 1);
            L.kg = n.h_pad / 16 + ab;
            L.last_steps = (a_last + 3) / 4;
        } else {
            L.kg = (L.k + 15) / 16;
            const int rem = L.k - 16 * (L.kg - 1);
            L.last_steps = rem < 4 ? rem : 4;
        }
        L.split = (l == L_REP1 || l == L_DYN1 || l == L_REW1 || l == L_POL1 || l == L_VAL1) ? 1 : 0;
        L.kq = (L.kg + 3) / 4;
    }
    MlpLds& o = p->o;
    int off = 0;
    const int xk = n.in_pad > n.x_pad ? n.in_pad : n.x_pad;
    o.X = off; off += xk * 16;
    o.H1 = off; off += n.p_pad * 16;
    o.V1 = off; off += n.p_pad * 16;
    o.HN = off; off += n.h_pad * 16;
    o.HS = off; off += n.h_pad * 16;
    int mx = n.A; mx = n.Sv > mx ? n.Sv : mx; mx = n.Sr > mx ? n.Sr : mx;
    o.lg_stride = pad16(mx) + 1;  // odd stride: 16 envs read their logits rows without LDS bank conflicts
    o.LG = off; off += ((2 * 16 * o.lg_stride + 3) & ~3);
    o.OUT = off; off += 64;
    o.BIAS = off;
    for (int l = 0; l < L_COUNT; l++) { n.L[l].b_lds = off; off += n.L[l].n_tiles * 16; }
    o.PM = off; off += 128;
    o.total_floats = off;

    // search kernel: tree part after the network part
    SearchParams& s = p->sp;
    s.net = n; s.o = o;
    s.S = c.num_simulations; s.A = c.num_actions; s.NN = c.num_simulations + 1;
    int b = o.total_floats * 4;
    auto take = [&](int bytes, int align) { b = (b + align - 1) / align * align; int r = b; b += bytes; return r; };
    s.t_nodes = take(16 * s.NN * (int)sizeof(TreeNode), 16);
    s.t_child = take(16 * s.NN * s.A * 2, 16);
    s.t_prior = take(16 * s.A * 8, 16);
    s.t_tmp = take(16 * s.A * 8, 16);
    s.t_pi0 = take(16 * s.A * 4, 16);
    s.t_mm = take(16 * 2 * 8, 16);
    s.t_sel = take(128 * 4, 16);
    s.t_ptr = take(32 * 8, 16);
    s.t_ftab = take((s.S + 1) * (s.S + 1) * 8, 16);
    s.lds_bytes = (b + 15) & ~15;
    // tree_mode 2 layout (mz_tree2.h) replaces t_nodes / t_child / t_ftab when it fits in LDS
    p->lds_mode0 = s.lds_bytes;
    p->tree2_ok = false;
    if (s.A <= 16 && s.NN < 255) {
        b = o.total_floats * 4;
        const int n2 = take(16 * s.NN * 16, 16), e2 = take(16 * s.NN * s.A * 16, 16), pr = take(16 * s.A * 8, 16), tm = take(16 * s.A * 8, 16),
                  p0 = take(16 * s.A * 4, 16), mmo = take(16 * 2 * 8, 16), se = take(128 * 4, 16), pt = take(32 * 8, 16),
                  ft = take(((s.S + 1) * (s.S + 2) / 2) * 8, 16), ca = take(16 * (s.NN + 1) * 16, 16), pa = take(16 * (s.NN + 3) * 2, 16),
                  ve = take(16 * 32, 16);
        const int total = (b + 15) & ~15;
        if (total <= 160 * 1024) {
            p->tree2_ok = true;
            p->lds_mode2 = total;
            p->sp2 = s;
            SearchParams& q = p->sp2;
            q.t2_nodes = n2; q.t2_entries = e2; q.t_prior = pr; q.t_tmp = tm; q.t_pi0 = p0; q.t_mm = mmo; q.t_sel = se; q.t_ptr = pt;
            q.t2_ftab = ft; q.t_cache = ca; q.t_path = pa; q.t_ver = ve;
            q.lds_bytes = total; q.tree_mode = 2;
        }
        constexpr int kFastPart = 4 * 4 * 256;  // floats of H1 / V1 the tuned kernel touches: [4 waves][<= 4 tiles] float4[64]
        if (!p->tree2_ok && n.p_pad * 16 > kFastPart) {
            const int delta = 2 * (n.p_pad * 16 - kFastPart);
            MlpLds of = o;
            of.V1 = o.H1 + kFastPart;
            of.HN -= delta; of.HS -= delta; of.LG -= delta; of.OUT -= delta; of.BIAS -= delta; of.PM -= delta; of.total_floats -= delta;
            b = of.total_floats * 4;
            const int n2 = take(16 * s.NN * 16, 16), e2 = take(16 * s.NN * s.A * 16, 16), pr = take(16 * s.A * 8, 16), tm = take(16 * s.A * 8, 16),
                      p0 = take(16 * s.A * 4, 16), mmo = take(16 * 2 * 8, 16), se = take(128 * 4, 16), pt = take(32 * 8, 16),
                      ft = take(((s.S + 1) * (s.S + 2) / 2) * 8, 16), ca = take(16 * (s.NN + 1) * 16, 16), pa = take(16 * (s.NN + 3) * 2, 16),
                      ve = take(16 * 32, 16);
            const int total = (b + 15) & ~15;
            if (total <= 160 * 1024) {
                p->tree2f_ok = true;
                p->lds_mode2f = total;
                p->fast_delta = delta;
                p->sp2f = s;
                SearchParams& q = p->sp2f;
                q.o = of;
                q.t2_nodes = n2; q.t2_entries = e2; q.t_prior = pr; q.t_tmp = tm; q.t_pi0 = p0; q.t_mm = mmo; q.t_sel = se; q.t_ptr = pt;
                q.t2_ftab = ft; q.t_cache = ca; q.t_path = pa; q.t_ver = ve;
                q.lds_bytes = total; q.tree_mode = 2;
            }
        }
    }

    InferParams& ip = p->ip;
    ip.net = n; ip.o = o;
    b = o.total_floats * 4;
    ip.t_ptr = take(32 * 8, 16);
    ip.t_pi = take(16 * n.A * 4, 16);
    ip.t_act = take(16 * 4, 16);
    ip.lds_bytes = (b + 15) & ~15;
}

// conv nets: the tree part only, laid out from offset 0 of a per-workgroup HBM region (same structure as tree_mode 0 in LDS)
static void compute_layout_conv(mz_planner* p, SearchParams* target = nullptr) {
    const mz_config& c = p->cfg;
    SearchParams& s = target ? *target : p->sp;
    s.S = c.num_simulations; s.A = c.num_actions; s.NN = c.num_simulations + 1;
    int b = 0;
    auto take = [&](int bytes, int align) { b = (b + align - 1) / align * align; int r = b; b += bytes; return r; };
    s.t_nodes = take(16 * s.NN * (int)sizeof(TreeNode), 16);
    s.t_child = take(16 * s.NN * s.A * 2, 16);
    s.t_prior = take(16 * s.A * 8, 16);
    s.t_tmp = take(16 * s.A * 8, 16);
    s.t_pi0 = take(16 * s.A * 4, 16);
    s.t_mm = take(16 * 2 * 8, 16);
    s.t_sel = take(128 * 4, 16);
    s.t_ptr = take(32 * 8, 16);
    s.t_ftab = take((s.S + 1) * (s.S + 1) * 8, 16);
    s.lds_bytes = (b + 255) & ~255;
    s.tree_mode = 0;
    if (!target) {
        p->lds_mode0 = s.lds_bytes;
        p->tree2_ok = false;
    }
}

static int planner_init(mz_planner* p, bool conv);
extern "C" int mz_planner_destroy(mz_planner* p);

extern "C" int mz_planner_create(const mz_config* cfg, int device_id, mz_planner** out) {
    if (!cfg || !out) return fail(MZ_E_INVALID, "null argument");
    const bool conv = cfg->net_kind == MZ_NET_BOARD || cfg->net_kind == MZ_NET_ATARI;
    if (cfg->net_kind != MZ_NET_MLP && !conv) return fail(MZ_E_INVALID, "unknown net_kind");
    if (cfg->num_actions < 1 || cfg->num_actions > 256) return fail(MZ_E_INVALID, "num_actions must be in [1, 256]");
    if (cfg->num_simulations < 1 || cfg->num_simulations > 4000) return fail(MZ_E_INVALID, "num_simulations out of range");
    if (conv) {
        if (cfg->obs_c < 1 || cfg->obs_h < 1 || cfg->obs_w < 1 || cfg->num_res_blocks < 0) return fail(MZ_E_INVALID, "bad conv network dimensions");
        if (cfg->net_kind == MZ_NET_ATARI && (cfg->obs_h != 96 || cfg->obs_w != 96))
            return fail(MZ_E_INVALID, "MuZeroAtariNet takes 96x96 frames (its hidden state is fixed at 6x6, network.py:515)");
        if (cfg->net_kind == MZ_NET_BOARD && cfg->obs_h * cfg->obs_w > 240) return fail(MZ_E_INVALID, "board larger than 240 points");
        if (cfg->num_planes > 512) return fail(MZ_E_INVALID, "conv nets: num_planes must be <= 512");
    }
    if ((!conv && cfg->hidden_dim < 1) || cfg->num_planes < 1 || cfg->num_envs < 1) return fail(MZ_E_INVALID, "bad network/env dimensions");
    if (cfg->value_support_size < 1 || cfg->reward_support_size < 1 || cfg->value_support_size > 1023 || cfg->reward_support_size > 1023)
        return fail(MZ_E_INVALID, "support sizes must be in [1, 1023]");
    if (cfg->is_board_game && cfg->discount != 1.0) return fail(MZ_E_INVALID, "board games require discount == 1.0 (mcts.py:349-350)");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(MZ_E_HIP, "no HIP device visible: the planner has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(MZ_E_INVALID, "device_id out of range");
    mz_planner* p = new mz_planner();
    p->cfg = *cfg;
    p->device = device_id;
    const int rc = planner_init(p, conv);  // every failure past this point releases what was allocated so far
    if (rc) {
        (void)mz_planner_destroy(p);
        return rc;
    }
    *out = p;
    return MZ_OK;
}

static int planner_init(mz_planner* p, bool conv) {
    const mz_config* cfg = &p->cfg;
    const int device_id = p->device;
    {
        const char* fg = getenv("MZ_FORCE_GENERIC");
        p->force_generic = fg && fg[0] == '1';
        const char* fe = getenv("MZ_FUSE_ENV");
        // default: fused -- one launch per lock-step move instead of four (temperature kernel, record copy, search, env step).  Round 1
        // kept the kernels separate for long moves because the env step ran single-lane behind four dependent global round trips at the
        // search kernel's tail; with its inputs requested before the play-policy phase (mz_env.h, cartpole_prefetch) the fused C2 move
        // is 727.7 us against 734.6 us (same box, same build)
        p->fuse_env = fe ? fe[0] != '0' : true;
        const char* gw = getenv("MZ_GTREE_WAVE");
        p->gtree_wave = gw ? gw[0] != '0' : cfg->num_actions <= 64 * MAX_CH64;  // (C5: +4.3 %; C4, six actions: +0.5 %)
        const char* hx = getenv("MZ_HWX");
        if (hx) p->hwx = atoi(hx);
        const char* to = getenv("MZ_TREE_OLD");
        p->tree_old = to && to[0] == '1';
        const char* nf = getenv("MZ_NO_FAST_LAYOUT");
        p->no_fast_layout = nf && nf[0] == '1';
        const char* a4 = getenv("MZ_FAST_AC4");
        p->fast_ac4 = !(a4 && a4[0] == '0');
    }
    p->conv = conv;
    if (conv) {
        ConvNetDev& n = p->cnet;
        n.kind = cfg->net_kind; n.in_c = cfg->obs_c; n.in_h = cfg->obs_h; n.in_w = cfg->obs_w; n.A = cfg->num_actions;
        n.R = cfg->num_res_blocks; n.P = cfg->num_planes; n.Sv = cfg->value_support_size; n.Sr = cfg->reward_support_size;
        n.hh = conv && cfg->net_kind == MZ_NET_ATARI ? 6 : cfg->obs_h;
        n.hw = conv && cfg->net_kind == MZ_NET_ATARI ? 6 : cfg->obs_w;
        p->cfg.hidden_dim = n.hidden_size();
        compute_layout_conv(p);
    } else {
        compute_layout(p);
    }
    if (!conv) {
        const char* ht = getenv("MZ_HBM_TREE");
        p->hbm_tree = (ht && ht[0] == '1') || cfg->num_actions > 16 * MAX_CH || (p->sp.lds_bytes > 160 * 1024 && !p->tree2_ok);
        if (p->hbm_tree) {
            if (p->ip.lds_bytes > 160 * 1024) {
                const int need = p->ip.lds_bytes;
                return fail(MZ_E_INVALID, "network needs " + std::to_string(need) + " bytes of LDS per workgroup (> 160 KiB)");
            }
            compute_layout_conv(p, &p->spg);
        }
    }
    HIPCHK(hipSetDevice(device_id));
    HIPCHK(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    const mz_config& c = p->cfg;
    const size_t B = (size_t)c.num_envs, A = (size_t)c.num_actions, S = (size_t)c.num_simulations;
    const int mt = c.max_ties > 0 ? c.max_ties : 4 * c.num_simulations + 8;
    p->cfg.max_ties = mt;
    HIPCHK(hipMalloc(&p->d_obs, B * obs_dim(c) * sizeof(float) + 256));  // + slack: conv staging reads whole 16-byte pixel quads
    HIPCHK(hipMalloc(&p->d_mask, B * A));
    HIPCHK(hipMalloc(&p->d_cur, B * sizeof(int)));
    HIPCHK(hipMalloc(&p->d_opp, B * sizeof(int)));
    HIPCHK(hipMalloc(&p->d_temp, B * sizeof(double)));
    HIPCHK(hipMalloc(&p->d_noise, B * A * sizeof(double)));
    HIPCHK(hipMalloc(&p->d_utie, B * (size_t)mt * sizeof(double)));
    HIPCHK(hipMalloc(&p->d_ufinal, B * sizeof(double)));
    HIPCHK(hipMalloc(&p->d_hidden, B * (S + 1) * (size_t)c.hidden_dim * sizeof(float) + 256));
    HIPCHK(hipMalloc(&p->d_ftab, (S + 1) * (S + 1) * sizeof(double)));
    HIPCHK(hipMalloc(&p->d_action, B * sizeof(int)));
    HIPCHK(hipMalloc(&p->d_pi, B * A * sizeof(double)));
    HIPCHK(hipMalloc(&p->d_root, B * sizeof(double)));
    HIPCHK(hipMalloc(&p->d_visits, B * A * sizeof(int)));
    HIPCHK(hipMalloc(&p->d_err, sizeof(int)));
    HIPCHK(hipMemset(p->d_err, 0, sizeof(int)));
    HIPCHK(hipMalloc(&p->d_stamps, 16 * sizeof(long long)));
    HIPCHK(hipMemset(p->d_stamps, 0, 16 * sizeof(long long)));
    HIPCHK(hipMemset(p->d_mask, 1, B * A));
    HIPCHK(hipDeviceSynchronize());  // (the fills above run on the NULL stream, which p->stream -- non-blocking -- never waits for)
    // child_U factor table: pb(N) / (n_child + 1) in float64 with the host libm, i.e. the very values math.log /
    // math.sqrt give the reference (mcts.py:193-195)
    std::vector<double> ft((S + 1) * (S + 1));
    for (size_t N = 0; N <= S; N++) {
        const double pb = (std::log(((double)N + c.pb_c_base + 1.0) / c.pb_c_base) + c.pb_c_init) * std::sqrt((double)N);
        for (size_t cn = 0; cn <= S; cn++) ft[N * (S + 1) + cn] = pb / (double)(cn + 1);
    }
    HIPCHK(hipMemcpy(p->d_ftab, ft.data(), ft.size() * sizeof(double), hipMemcpyHostToDevice));
    {
        std::vector<double> tri;
        for (size_t N = 0; N <= S; N++)
            for (size_t cn = 0; cn <= N; cn++) tri.push_back(ft[N * (S + 1) + cn]);
        HIPCHK(hipMalloc(&p->d_ftab_tri, tri.size() * sizeof(double)));
        HIPCHK(hipMemcpy(p->d_ftab_tri, tri.data(), tri.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    HIPCHK(hipEventCreate(&p->ev_begin));
    HIPCHK(hipEventCreate(&p->ev_end));
    if (conv || p->hbm_tree) {
        const size_t blocks = (B + TILE_E - 1) / TILE_E, HS = (size_t)p->cfg.hidden_dim;
        HIPCHK(hipMalloc(&p->d_regions, blocks * (size_t)(conv ? p->sp.lds_bytes : p->spg.lds_bytes)));
        HIPCHK(hipMalloc(&p->d_pi_scratch, B * A * sizeof(float)));
        HIPCHK(hipMalloc(&p->d_pi0, B * A * sizeof(float)));
        HIPCHK(hipMalloc(&p->d_sim_reward, B * sizeof(float)));
        HIPCHK(hipMalloc(&p->d_sim_value, B * sizeof(float)));
        HIPCHK(hipMalloc(&p->d_srcptrs, B * sizeof(float*)));
        HIPCHK(hipMalloc(&p->d_dstptrs, B * sizeof(float*)));
        HIPCHK(hipMalloc(&p->d_rootptrs, B * sizeof(float*)));
        HIPCHK(hipMalloc(&p->d_sim_action, B * sizeof(int)));
        std::vector<float*> roots(B);
        for (size_t i = 0; i < B; i++) roots[i] = p->d_hidden + i * (S + 1) * HS;
        HIPCHK(hipMemcpy(p->d_rootptrs, roots.data(), B * sizeof(float*), hipMemcpyHostToDevice));
        if (conv) {
            return MZ_OK;
        }
    }
    int max_lds = p->tree2_ok ? p->lds_mode2 : (p->tree2f_ok ? p->lds_mode2f : 0);
    if (p->lds_mode0 <= 160 * 1024 && p->lds_mode0 > max_lds) max_lds = p->lds_mode0;
    if (max_lds == 0) max_lds = p->ip.lds_bytes;  // hbm_tree: the LDS-resident search kernels are never launched
    if (p->lds_mode0 > 160 * 1024) p->tree_old = false;  // only the mode-2 layout fits
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search<false>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search<true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_infer<false>), hipFuncAttributeMaxDynamicSharedMemorySize, p->ip.lds_bytes));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_infer<true>), hipFuncAttributeMaxDynamicSharedMemorySize, p->ip.lds_bytes));
    if (c.hidden_dim == 64 && (c.num_planes == 256 || c.num_planes == 512) && c.num_actions <= 16 && c.value_support_size <= 32 &&
        c.reward_support_size <= 32 && (c.value_support_size + 15) / 16 == (c.reward_support_size + 15) / 16 &&
        (size_t)c.num_envs * (c.num_simulations + 1) * 256 < ((size_t)1 << 32)) {  // (the tuned kernel addresses the node store with 32-bit byte offsets)
        p->fast_planes = c.num_planes;
#define MZ_FAST_LDS(PL, T, F, W) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search_fast<PL, T, T, F, W, kFastHW>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds))
#define MZ_FAST_LDS4(PL, T) MZ_FAST_LDS(PL, T, false, 0); MZ_FAST_LDS(PL, T, false, 2); MZ_FAST_LDS(PL, T, true, 0); MZ_FAST_LDS(PL, T, true, 2)
        MZ_FAST_LDS(256, 1, false, 10); MZ_FAST_LDS(256, 1, true, 10);  // ten actions (TicTacToe)
        MZ_FAST_LDS(512, 2, false, 4); MZ_FAST_LDS(512, 2, true, 4);    // four actions, the classic-control net (LunarLander's shape)
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search_fast<256, 1, 1, false, 10, kFastHW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search_fast<256, 1, 1, true, 10, kFastHW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
#ifdef MZ_DEV_SHAPES  // development builds: only the C2 / C3 shapes of the tuned kernel (a third of the compile time)
        MZ_FAST_LDS4(256, 1); MZ_FAST_LDS4(512, 2);
#else
        MZ_FAST_LDS4(256, 1); MZ_FAST_LDS4(256, 2); MZ_FAST_LDS4(512, 1); MZ_FAST_LDS4(512, 2);
#endif
#undef MZ_FAST_LDS4
#undef MZ_FAST_LDS
    }
    return MZ_OK;
}

extern "C" int mz_planner_destroy(mz_planner* p) {
    if (!p) return MZ_OK;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    void* bufs[] = {p->d_obs, p->d_mask, p->d_cur, p->d_opp, p->d_temp, p->d_noise, p->d_utie, p->d_ufinal, p->d_hidden, p->d_ftab, p->d_ftab_tri,
                    p->d_action, p->d_pi, p->d_root, p->d_visits, p->d_err, p->d_stamps, p->d_spi0, p->d_svalues, p->d_srewards, p->d_tparent,
                    p->d_taction, p->d_inf_in, p->d_inf_hidden, p->d_inf_reward, p->d_inf_value, p->d_inf_pi, p->d_inf_action};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    for (int l = 0; l < L_COUNT; l++) {
        if (p->d_w[l]) (void)hipFree(p->d_w[l]);
        if (p->d_b[l]) (void)hipFree(p->d_b[l]);
    }
    if (p->d_stream[0]) (void)hipFree(p->d_stream[0]);
    if (p->d_bias_all) (void)hipFree(p->d_bias_all);
    if (p->d_dbg_noise) { (void)hipFree(p->d_dbg_noise); (void)hipFree(p->d_dbg_utie); (void)hipFree(p->d_dbg_ufinal); }
    if (p->d_epi_ctr) (void)hipFree(p->d_epi_ctr);
    if (p->d_epi_off) (void)hipFree(p->d_epi_off);
    void* cbufs[] = {p->d_pi_scratch, p->d_regions, p->d_pi0, p->d_sim_reward, p->d_sim_value, (void*)p->d_srcptrs, p->d_dstptrs, p->d_rootptrs, p->d_sim_action};
    for (void* b : cbufs)
        if (b) (void)hipFree(b);
    convnet_free(p->cnet);
    env_free(p->env);
    for (auto& pr : p->kev) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (p->ev_begin) (void)hipEventDestroy(p->ev_begin);
    if (p->ev_end) (void)hipEventDestroy(p->ev_end);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return MZ_OK;
}

extern "C" int32_t mz_planner_hidden_size(const mz_planner* p) { return p ? p->cfg.hidden_dim : 0; }

extern "C" int mz_planner_set_param(mz_planner* p, const char* name, const float* h_data, const int64_t* shape, int32_t ndim) {
    if (!p || !name || !h_data || !shape || ndim < 1 || ndim > 4) return fail(MZ_E_INVALID, "bad argument to mz_planner_set_param");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; i++) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(h_data, h_data + n);
    p->params[name] = std::move(t);
    p->committed = false;
    return MZ_OK;
}

// column of layer l's weight matrix that lane group q supplies for k-step s of block g (-1: padding).  Hidden-type
// blocks: 16g + 4q + s; the dynamics layer's action block(s): action 16(g - hblocks) + 4s + q (mz_mlp.h header).
static int packed_k(const MlpNet& n, int l, int g, int q, int s) {
    const MlpLayer& L = n.L[l];
    if (l == L_DYN0) {
        const int hb = n.h_pad / 16;
        if (g >= hb) {
            const int a = 16 * (g - hb) + 4 * s + q;
            return a < n.A ? n.H + a : -1;
        }
        const int k = 16 * g + 4 * q + s;
        return k < n.H ? k : -1;
    }
    const int k = 16 * g + 4 * q + s;
    return k < L.k ? k : -1;
}

// the ten-action instantiation of the tuned kernel: ten actions AND MSE heads (TicTacToe's MLP net, config.py:106-136)
static bool fast_ac10(const mz_planner* p) {
    const mz_config& c = p->cfg;
    return p->fast_planes == 256 && c.num_actions == 10 && c.value_support_size == 1 && c.reward_support_size == 1;
}

// the two-action instantiations: two actions, single player, categorical reward and value heads (classic control)
static bool fast_two_act(const mz_config& c) {
    return c.num_actions == 2 && !c.is_board_game && c.reward_support_size > 1 && c.value_support_size > 1;
}

extern "C" int mz_planner_commit_params(mz_planner* p) {
    if (!p) return fail(MZ_E_INVALID, "null planner");
    HIPCHK(hipSetDevice(p->device));
    if (p->conv) {
        HIPCHK(hipStreamSynchronize(p->stream));
        ConvNetDev fresh = p->cnet;
        fresh.allocs.clear(); fresh.rep_res.clear(); fresh.dyn_res.clear(); fresh.pred_res.clear();
        fresh.bufA = fresh.bufB = fresh.bufC = nullptr; fresh.buf_elems = 0;
        fresh.tw_rep = fresh.tb_rep = fresh.tw_dyn = fresh.tb_dyn = fresh.tw_pred = fresh.tb_pred = nullptr;
        fresh.dyn_act_w = nullptr; fresh.dyn_inv_hw = 0; fresh.dyn_sp_w = nullptr; fresh.dyn_sp_terms = nullptr;
        convnet_free(p->cnet);
        p->cnet = fresh;
        ParamMap pm;
        for (auto& kv : p->params) pm[kv.first] = HostTensorRef{kv.second.data.data(), kv.second.shape};
        std::string err;
        const int rc = convnet_build(p->cnet, pm, &err);
        if (rc) return fail(rc == -2 ? MZ_E_HIP : (err.rfind("missing", 0) == 0 ? MZ_E_STATE : MZ_E_INVALID), err);
        hipError_t e = convnet_ensure_buffers(p->cnet, p->cfg.num_envs);
        if (e != hipSuccess) return fail(MZ_E_HIP, std::string("conv work buffers: ") + hipGetErrorString(e));
        p->committed = true;
        return MZ_OK;
    }
    for (int l = 0; l < L_COUNT; l++) {
        const MlpLayer& L = p->net.L[l];
        const std::string wn = std::string(kMlpNames[l]) + ".weight", bn = std::string(kMlpNames[l]) + ".bias";
        auto wi = p->params.find(wn), bi = p->params.find(bn);
        if (wi == p->params.end() || bi == p->params.end()) return fail(MZ_E_STATE, "missing parameter " + wn + " / " + bn);
        const HostTensor &W = wi->second, &Bv = bi->second;
        if (W.shape.size() != 2 || W.shape[0] != L.n || W.shape[1] != L.k || Bv.data.size() != (size_t)L.n)
            return fail(MZ_E_INVALID, "shape mismatch for " + wn + ": expected [" + std::to_string(L.n) + ", " + std::to_string(L.k) + "]");
        // A-operand fragment order of v_mfma_f32_16x16x4_f32 in the summation order of mz_mlp.h
        std::vector<float> pw((size_t)L.n_tiles * L.kg * 256, 0.0f), pb((size_t)L.n_tiles * 16, 0.0f);
        for (int t = 0; t < L.n_tiles; t++)
            for (int g = 0; g < L.kg; g++)
                for (int lane = 0; lane < 64; lane++)
                    for (int s = 0; s < 4; s++) {
                        const int nn = 16 * t + (lane & 15), kk = packed_k(p->net, l, g, lane >> 4, s);
                        if (nn < L.n && kk >= 0) pw[(((size_t)t * L.kg + g) * 64 + lane) * 4 + s] = W.data[(size_t)nn * L.k + kk];
                    }
        for (int i = 0; i < L.n; i++) pb[i] = Bv.data[i];
        if (!p->d_w[l]) HIPCHK(hipMalloc(&p->d_w[l], pw.size() * sizeof(float)));
        if (!p->d_b[l]) HIPCHK(hipMalloc(&p->d_b[l], pb.size() * sizeof(float)));
        HIPCHK(hipMemcpy(p->d_w[l], pw.data(), pw.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(p->d_b[l], pb.data(), pb.size() * sizeof(float), hipMemcpyHostToDevice));
        p->net.L[l].w = p->d_w[l];
        p->net.L[l].b = p->d_b[l];
    }
    {
        // all padded bias vectors in one buffer, laid out like their LDS copies (stage_biases)
        const int base = p->o.BIAS, count = p->o.PM - p->o.BIAS;
        std::vector<float> all((size_t)count, 0.0f);
        for (int l = 0; l < L_COUNT; l++) {
            const HostTensor& Bv = p->params.find(std::string(kMlpNames[l]) + ".bias")->second;
            for (int i = 0; i < p->net.L[l].n; i++) all[p->net.L[l].b_lds - base + i] = Bv.data[i];
        }
        if (!p->d_bias_all) HIPCHK(hipMalloc(&p->d_bias_all, all.size() * sizeof(float)));
        HIPCHK(hipMemcpy(p->d_bias_all, all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice));
        p->net.b_all = p->d_bias_all; p->net.b_base = base; p->net.b_count = count;
    }
    p->sp.net = p->net;
    p->ip.net = p->net;
    if (p->fast_planes) {
        // ONE per-wave weight stream in consumption order (layout: mz_search_fast.h header)
        const bool sc = kFastSC && fast_ac10(p), ax = kFastAX && fast_ac10(p);  // (mz_search_fast.h: scalar heads / action column outside the stream)
        const int NT = p->fast_planes / 64, TR = sc ? 0 : p->net.L[L_REW1].n_tiles, TV = sc ? 0 : p->net.L[L_VAL1].n_tiles, RD = fast_rd(p->fast_planes, fast_two_act(p->cfg) ? 2 : 0);
        const int XG = ax ? 4 : 5;
        const int I_D1 = 0, I_D2 = I_D1 + XG, I_R1 = I_D2 + 4, I_R2 = I_R1 + 4, I_V1 = I_R2 + TR, I_V2 = I_V1 + 4, I_END = I_V2 + TV;
        const int SL = (I_END + RD - 1) / RD * RD;
        const size_t stream_floats = (size_t)WG_WAVES * SL * NT * 256;
        std::vector<float> st(stream_floats + (ax ? (size_t)p->cfg.num_actions * p->fast_planes : 0), 0.0f);
        auto put = [&](int w, int slot, int j, int l, int row_tile, int g) {
            const MlpLayer& L = p->net.L[l];
            const HostTensor& W = p->params.find(std::string(kMlpNames[l]) + ".weight")->second;
            float* d = &st[(((size_t)w * SL + slot) * NT + j) * 256];
            for (int lane = 0; lane < 64; lane++)
                for (int i = 0; i < 4; i++) {
                    const int nn = 16 * row_tile + (lane & 15), kk = packed_k(p->net, l, g, lane >> 4, i);
                    if (nn < L.n && kk >= 0) d[lane * 4 + i] = W.data[(size_t)nn * L.k + kk];
                }
        };
        for (int w = 0; w < WG_WAVES; w++) {
            for (int g = 0; g < XG; g++)
                for (int j = 0; j < NT; j++) put(w, I_D1 + g, j, L_DYN0, NT * w + j, g);
            for (int idx = 0; idx < 4 * NT; idx++) put(w, I_D2 + idx / NT, idx % NT, L_DYN1, idx % 4, NT * w + idx / 4);
            for (int g = 0; g < 4; g++)
                for (int j = 0; j < NT; j++) put(w, I_R1 + g, j, L_REW0, NT * w + j, g);
            for (int idx = 0; idx < TR * NT; idx++) put(w, I_R2 + idx / NT, idx % NT, L_REW1, idx % TR, NT * w + idx / TR);
            for (int g = 0; g < 4; g++)
                for (int j = 0; j < NT; j++) put(w, I_V1 + g, j, L_VAL0, NT * w + j, g);
            for (int idx = 0; idx < TV * NT; idx++) put(w, I_V2 + idx / NT, idx % NT, L_VAL1, idx % TV, NT * w + idx / TV);
        }
        if (ax) {  // the action columns of the dynamics net's first layer, one row per action
            const HostTensor& W = p->params.find(std::string(kMlpNames[L_DYN0]) + ".weight")->second;
            const MlpLayer& L = p->net.L[L_DYN0];
            for (int a = 0; a < p->cfg.num_actions; a++)
                for (int r = 0; r < L.n; r++) st[stream_floats + (size_t)a * p->fast_planes + r] = W.data[(size_t)r * L.k + p->net.H + a];
        }
        if (!p->d_stream[0]) HIPCHK(hipMalloc(&p->d_stream[0], st.size() * sizeof(float)));
        HIPCHK(hipMemcpy(p->d_stream[0], st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
        p->fw.stream = reinterpret_cast<const float4*>(p->d_stream[0]);
        p->fw.bytes = (unsigned)(stream_floats * sizeof(float));
        p->fw.wact = ax ? p->d_stream[0] + stream_floats : nullptr;
    }
    p->committed = true;
    return MZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// inference API
// ---------------------------------------------------------------------------------------------------------
static int ensure_infer_buffers(mz_planner* p, int batch) {
    if (batch <= p->inf_cap) return MZ_OK;
    void* old[] = {p->d_inf_in, p->d_inf_hidden, p->d_inf_reward, p->d_inf_value, p->d_inf_pi, p->d_inf_action};
    for (void* b : old)
        if (b) (void)hipFree(b);
    const size_t B = (size_t)batch;
    const size_t in = (size_t)(obs_dim(p->cfg) > p->cfg.hidden_dim ? obs_dim(p->cfg) : p->cfg.hidden_dim);
    HIPCHK(hipMalloc(&p->d_inf_in, B * in * sizeof(float) + 256));
    HIPCHK(hipMalloc(&p->d_inf_hidden, B * p->cfg.hidden_dim * sizeof(float)));
    HIPCHK(hipMalloc(&p->d_inf_reward, B * sizeof(float)));
    HIPCHK(hipMalloc(&p->d_inf_value, B * sizeof(float)));
    HIPCHK(hipMalloc(&p->d_inf_pi, B * p->cfg.num_actions * sizeof(float)));
    HIPCHK(hipMalloc(&p->d_inf_action, B * sizeof(int)));
    p->inf_cap = batch;
    return MZ_OK;
}

static int run_infer(mz_planner* p, bool initial, int batch, const float* h_in, const int32_t* h_action, float* h_hidden, float* h_reward,
                     float* h_pi, float* h_value) {
    if (!p || batch < 1 || !h_in) return fail(MZ_E_INVALID, "bad argument to inference call");
    if (!p->committed) return fail(MZ_E_STATE, "weights not committed: call mz_planner_set_param for every tensor, then mz_planner_commit_params");
    HIPCHK(hipSetDevice(p->device));
    int rc = ensure_infer_buffers(p, batch);
    if (rc) return rc;
    const size_t in_w = initial ? obs_dim(p->cfg) : p->cfg.hidden_dim;
    HIPCHK(hipMemcpyAsync(p->d_inf_in, h_in, (size_t)batch * in_w * sizeof(float), hipMemcpyHostToDevice, p->stream));
    if (!initial) HIPCHK(hipMemcpyAsync(p->d_inf_action, h_action, (size_t)batch * sizeof(int), hipMemcpyHostToDevice, p->stream));
    if (p->conv) {
        hipError_t e = convnet_ensure_buffers(p->cnet, batch);
        if (e != hipSuccess) return fail(MZ_E_HIP, std::string("conv work buffers: ") + hipGetErrorString(e));
        if (initial) convnet_initial(p->stream, p->cnet, batch, p->d_inf_in, nullptr, p->d_inf_hidden, p->d_inf_pi, p->d_inf_value);
        else convnet_recurrent(p->stream, p->cnet, batch, nullptr, p->d_inf_in, p->d_inf_action, nullptr, p->d_inf_hidden, p->d_inf_reward,
                               p->d_inf_value, p->d_inf_pi);
        HIPCHK(hipGetLastError());
    } else {
    InferParams ip = p->ip;
    ip.B = batch; ip.in = p->d_inf_in; ip.in_ptrs = nullptr; ip.out_ptrs = nullptr; ip.action = p->d_inf_action; ip.hidden_out = p->d_inf_hidden; ip.reward = p->d_inf_reward;
    ip.value = p->d_inf_value; ip.pi = p->d_inf_pi;
    const dim3 grid((batch + TILE_E - 1) / TILE_E), block(WG_THREADS);
    if (initial) hipLaunchKernelGGL(k_infer<true>, grid, block, ip.lds_bytes, p->stream, ip);
    else hipLaunchKernelGGL(k_infer<false>, grid, block, ip.lds_bytes, p->stream, ip);
    HIPCHK(hipGetLastError());
    }
    if (h_hidden) HIPCHK(hipMemcpyAsync(h_hidden, p->d_inf_hidden, (size_t)batch * p->cfg.hidden_dim * sizeof(float), hipMemcpyDeviceToHost, p->stream));
    if (h_reward) HIPCHK(hipMemcpyAsync(h_reward, p->d_inf_reward, (size_t)batch * sizeof(float), hipMemcpyDeviceToHost, p->stream));
    if (h_value) HIPCHK(hipMemcpyAsync(h_value, p->d_inf_value, (size_t)batch * sizeof(float), hipMemcpyDeviceToHost, p->stream));
    if (h_pi) HIPCHK(hipMemcpyAsync(h_pi, p->d_inf_pi, (size_t)batch * p->cfg.num_actions * sizeof(float), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    return MZ_OK;
}

extern "C" int mz_planner_initial_inference(mz_planner* p, int32_t batch, const float* h_obs, float* h_hidden, float* h_pi, float* h_value) {
    return run_infer(p, true, batch, h_obs, nullptr, h_hidden, nullptr, h_pi, h_value);
}

extern "C" int mz_planner_recurrent_inference(mz_planner* p, int32_t batch, const float* h_hidden, const int32_t* h_action, float* h_hidden_out,
                                              float* h_reward, float* h_pi, float* h_value) {
    if (!h_action) return fail(MZ_E_INVALID, "null action");
    return run_infer(p, false, batch, h_hidden, h_action, h_hidden_out, h_reward, h_pi, h_value);
}

// ---------------------------------------------------------------------------------------------------------
// search
// ---------------------------------------------------------------------------------------------------------
static int next_kernel_events(mz_planner* p, hipEvent_t* a, hipEvent_t* b) {
    if (p->kev_used == p->kev.size()) {
        hipEvent_t x, y;
        HIPCHK(hipEventCreate(&x));
        HIPCHK(hipEventCreate(&y));
        p->kev.emplace_back(x, y);
    }
    *a = p->kev[p->kev_used].first;
    *b = p->kev[p->kev_used].second;
    p->kev_used++;
    return MZ_OK;
}

// Whether the dispatch of launch_search has a k_search_fast build for this net's categorical-head shape.  Always true in the product build;
// -DMZ_DEV_SHAPES compiles only two of the four (planes, head tiles) builds and falls back to the shape-generic kernel for the others, which
// must then keep the generic LDS carve-out (the fast carve-out under the generic kernel overruns LDS).
static bool fast_build_exists(const mz_planner* p) {
#ifdef MZ_DEV_SHAPES
    const int two = p->net.L[L_VAL1].n_tiles == 2;
    const mz_config& c = p->cfg;
    const bool four_act = p->fast_ac4 && c.num_actions == 4 && c.reward_support_size > 1 && c.value_support_size > 1 && p->fast_planes == 512 && two;
    return four_act || (p->fast_planes == 512 && two) || (p->fast_planes == 256 && !two);
#else
    (void)p;
    return true;
#endif
}

// launches the fused search kernel over inputs that are already resident in the planner's device buffers
static int launch_search(mz_planner* p, int batch, int deterministic, bool has_mask, bool injected_rng, bool scripted, const EnvLaunch* fenv = nullptr) {
    const mz_config& c = p->cfg;
    // the tuned kernel with its own LDS carve-out (sp2f): only where the generic carve-out does not fit AND k_search_fast is the kernel
    // the dispatch below picks (categorical heads; the ten-action MSE build is 256 planes wide and always fits the generic one)
    const bool fastlayout = !p->tree2_ok && p->tree2f_ok && p->fast_planes && !p->force_generic && !p->tree_old && !scripted &&
                            c.value_support_size > 1 && c.reward_support_size > 1 && !p->no_fast_layout && fast_build_exists(p);
    const bool mode2 = (p->tree2_ok && !p->tree_old) || fastlayout;
    SearchParams s = fastlayout ? p->sp2f : (mode2 ? p->sp2 : p->sp);
    s.tree_mode = mode2 ? 2 : 0;
    s.net = p->net;
    if (fastlayout) {  // the bias block sits `fast_delta` floats lower in this carve-out
        s.net.b_base -= p->fast_delta;
        for (int l = 0; l < L_COUNT; l++) s.net.L[l].b_lds -= p->fast_delta;
    }
    s.ftab_tri = p->d_ftab_tri;
    s.discount = c.discount; s.board = c.is_board_game; s.has_bounds = c.has_known_bounds;
    s.kb_min = c.known_bounds_min; s.kb_max = c.known_bounds_max; s.alpha = c.root_dirichlet_alpha; s.eps = c.root_exploration_eps;
    s.deterministic = deterministic; s.has_mask = has_mask ? 1 : 0;
    const bool want_noise = !deterministic && c.root_dirichlet_alpha > 0.0 && c.root_exploration_eps > 0.0;  // mcts.py:361
    s.noise_mode = want_noise ? (injected_rng ? 1 : 2) : 0; s.legacy_promo = c.legacy_scalar_promotion ? 1 : 0;
    s.rng_mode = injected_rng ? 0 : 1;
    s.max_ties = c.max_ties;
    s.B = batch;
    s.obs = p->d_obs; s.mask = p->d_mask; s.cur = p->d_cur; s.opp = p->d_opp; s.temperature = p->d_temp;
    s.noise = p->d_noise; s.u_tie = p->d_utie; s.u_final = p->d_ufinal; s.hidden = p->d_hidden; s.ftab = p->d_ftab;
    s.out_action = p->d_action; s.out_pi = p->d_pi; s.out_root = p->d_root; s.out_visits = p->d_visits; s.err = p->d_err;
    s.s_pi0 = p->d_spi0; s.s_values = p->d_svalues; s.s_rewards = p->d_srewards; s.trace_parent = p->d_tparent; s.trace_action = p->d_taction;
    s.seed = c.seed; s.move_counter = p->move_counter++; s.env_offset = 0; s.stamps = p->d_stamps;
    s.dbg_noise = p->d_dbg_noise; s.dbg_utie = p->d_dbg_utie; s.dbg_ufinal = p->d_dbg_ufinal;
    s.fuse_env = fenv ? 1 : 0;
    // both jobs where both heads are categorical (classic control: -3.8 % on C2, same box); the normalisation alone for the MSE heads of
    // the board games, which have no softmax row (C3: -1.2 %)
    s.hwx = p->hwx >= 0 ? p->hwx : ((c.reward_support_size > 1 && c.value_support_size > 1) ? 3 : 1);
    if (fenv) s.fenv = *fenv;
    const dim3 grid((batch + TILE_E - 1) / TILE_E), block(WG_THREADS);
    hipEvent_t ea = nullptr, eb = nullptr;
    if (p->profiling) {
        int rc = next_kernel_events(p, &ea, &eb);
        if (rc) return rc;
        HIPCHK(hipEventRecord(ea, p->stream));
    }
    auto fast_name = [&](int planes, int tiles, bool fuse, int ac, bool spb) {
        char b[160];
        snprintf(b, sizeof b, "k_search_fast<planes=%d, TR=%d, TV=%d, FUSE=%s, AC=%d, HW=%s%s> (LDS trees%s, one launch per move)", planes, tiles, tiles,
                 fuse ? "true" : "false", ac, kFastHW ? "true" : "false", spb ? ", SPB=true" : "", fastlayout ? ", own carve-out" : "");
        p->last_dispatch = b;
    };
    if (p->conv || p->hbm_tree) {
        // HBM-resident trees: root inference -> init -> S x {select, network evaluation, expand + backup} -> play
        const bool mlp = !p->conv;
        p->last_dispatch = std::string(mlp ? "k_infer" : "conv towers (mz_convnet.h: k_conv3x3 / k_res_tower / k_head)") + " around HBM trees: " +
                           (p->gtree_wave ? "k_gtree_select_wave" : "k_gtree_select") + " + k_gtree_backup per simulation";
        InferParams ip = p->ip;
        ip.net = p->net; ip.B = batch; ip.in = nullptr; ip.in_ptrs = nullptr; ip.action = p->d_sim_action; ip.hidden_out = nullptr;
        ip.out_ptrs = nullptr; ip.reward = p->d_sim_reward; ip.value = p->d_sim_value; ip.pi = p->d_pi_scratch;
        s = mlp ? p->spg : p->sp;
        s.tree_mode = 0;
        s.discount = c.discount; s.board = c.is_board_game; s.has_bounds = c.has_known_bounds;
        s.kb_min = c.known_bounds_min; s.kb_max = c.known_bounds_max; s.alpha = c.root_dirichlet_alpha; s.eps = c.root_exploration_eps;
        s.deterministic = deterministic; s.has_mask = has_mask ? 1 : 0;
        s.noise_mode = want_noise ? (injected_rng ? 1 : 2) : 0; s.legacy_promo = c.legacy_scalar_promotion ? 1 : 0;
        s.rng_mode = injected_rng ? 0 : 1;
        s.max_ties = c.max_ties; s.B = batch;
        s.obs = p->d_obs; s.mask = p->d_mask; s.cur = p->d_cur; s.opp = p->d_opp; s.temperature = p->d_temp;
        s.noise = p->d_noise; s.u_tie = p->d_utie; s.u_final = p->d_ufinal; s.hidden = p->d_hidden; s.ftab = p->d_ftab;
        s.out_action = p->d_action; s.out_pi = p->d_pi; s.out_root = p->d_root; s.out_visits = p->d_visits; s.err = p->d_err;
        s.trace_parent = scripted ? p->d_tparent : nullptr; s.trace_action = scripted ? p->d_taction : nullptr;
        s.seed = c.seed; s.move_counter = p->move_counter - 1; s.env_offset = 0; s.stamps = nullptr;
        s.dbg_noise = p->d_dbg_noise; s.dbg_utie = p->d_dbg_utie; s.dbg_ufinal = p->d_dbg_ufinal;
        GTreeLaunch G{};
        G.P = s; G.regions = p->d_regions; G.hidden_size = c.hidden_dim; G.src_ptrs = p->d_srcptrs; G.dst_ptrs = p->d_dstptrs;
        G.actions = p->d_sim_action;
        if (scripted) {
            G.pi0 = p->d_spi0;
        } else if (mlp) {
            InferParams r = ip;
            r.in = p->d_obs; r.out_ptrs = p->d_rootptrs; r.pi = p->d_pi0;
            hipLaunchKernelGGL(k_infer<true>, grid, block, r.lds_bytes, p->stream, r);  // root value discarded (mcts.py:356-367)
            G.pi0 = p->d_pi0;
        } else {
            convnet_initial(p->stream, p->cnet, batch, p->d_obs, p->d_rootptrs, nullptr, p->d_pi0, p->d_sim_value);  // root value discarded
            G.pi0 = p->d_pi0;
        }
        hipLaunchKernelGGL(k_gtree_init, grid, block, 0, p->stream, G);
        for (int sim = 0; sim < c.num_simulations; sim++) {
            G.sim = sim;
            // one wave per env (see k_gtree_select_wave); MZ_GTREE_WAVE=0/1 overrides (A/B measurements, tests)
            if (p->gtree_wave) hipLaunchKernelGGL(k_gtree_select_wave, dim3((batch + 3) / 4), block, 0, p->stream, G);
            else hipLaunchKernelGGL(k_gtree_select, grid, block, 0, p->stream, G);
            if (scripted) {
                G.reward = p->d_srewards + sim; G.value = p->d_svalues + sim; G.rv_stride = c.num_simulations;
            } else if (mlp) {
                InferParams r = ip;
                r.in_ptrs = p->d_srcptrs; r.out_ptrs = p->d_dstptrs;
                hipLaunchKernelGGL(k_infer<false>, grid, block, r.lds_bytes, p->stream, r);
                G.reward = p->d_sim_reward; G.value = p->d_sim_value; G.rv_stride = 1;
            } else {
                convnet_recurrent(p->stream, p->cnet, batch, p->d_srcptrs, nullptr, p->d_sim_action, p->d_dstptrs, nullptr, p->d_sim_reward,
                                  p->d_sim_value, nullptr, p->d_hidden, (size_t)c.num_envs * (c.num_simulations + 1) * (size_t)c.hidden_dim);
                G.reward = p->d_sim_reward; G.value = p->d_sim_value; G.rv_stride = 1;
            }
            hipLaunchKernelGGL(k_gtree_backup, grid, block, 0, p->stream, G);
        }
        hipLaunchKernelGGL(k_gtree_finish, grid, block, 0, p->stream, G);
    } else
    if (scripted) {
        p->last_dispatch = "k_search<SCRIPTED=true> (scripted-network test hook)";
        hipLaunchKernelGGL(k_search<true>, grid, block, s.lds_bytes, p->stream, s);
    } else if (p->fast_planes && !p->force_generic && s.tree_mode == 2) {  // (k_search_fast is written for the tree_mode 2 layout)
        const int two = p->net.L[L_VAL1].n_tiles == 2;
#define MZ_FAST4(PL, T, F, W) fast_name(PL, T, F, W, false); hipLaunchKernelGGL((k_search_fast<PL, T, T, F, W, kFastHW>), grid, dim3(kFastHW ? 2 * WG_THREADS : WG_THREADS), s.lds_bytes, p->stream, s, p->fw)
#define MZ_FAST(PL, T) do { if (fenv) { if (two_act) { MZ_FAST4(PL, T, true, 2); } else { MZ_FAST4(PL, T, true, 0); } } \
                            else { if (two_act) { MZ_FAST4(PL, T, false, 2); } else { MZ_FAST4(PL, T, false, 0); } } } while (0)
        // compile-time specialisation (mz_tree2.h, AM): two actions, single player, categorical reward and value heads
        const bool two_act = fast_two_act(c);  // (also fixes the depth of the weight ring the stream is packed for: load_weights)
        // four actions, categorical heads, the 512-plane net (mz_tree2.h, ACT: the backup's refresh unrolled, the action count a constant)
        const bool four_act = p->fast_ac4 && c.num_actions == 4 && c.reward_support_size > 1 && c.value_support_size > 1 && p->fast_planes == 512 && two;
        if (fast_ac10(p)) {  // (TicTacToe: ten actions)
            // SPB: the build with the board games' self-play settings as compile-time constants (mz_search_fast.h)
            const bool spb = s.board && s.has_bounds && s.discount == 1.0 && s.noise_mode == 2 && s.rng_mode == 1 && !s.deterministic && s.has_mask;
            const dim3 fblock(kFastHW ? 2 * WG_THREADS : WG_THREADS);
            if (spb) {
                fast_name(256, 1, fenv != nullptr, 10, true);
                if (fenv) hipLaunchKernelGGL((k_search_fast<256, 1, 1, true, 10, kFastHW, true>), grid, fblock, s.lds_bytes, p->stream, s, p->fw);
                else hipLaunchKernelGGL((k_search_fast<256, 1, 1, false, 10, kFastHW, true>), grid, fblock, s.lds_bytes, p->stream, s, p->fw);
            } else if (fenv) { MZ_FAST4(256, 1, true, 10); } else { MZ_FAST4(256, 1, false, 10); }
        } else
        if (four_act) { if (fenv) { MZ_FAST4(512, 2, true, 4); } else { MZ_FAST4(512, 2, false, 4); } }
        else if (c.value_support_size == 1 || c.reward_support_size == 1) {
            // an MSE head's one-neuron layer runs on the vector ALUs in its own summation order (mz_mlp.h, scalar_head_tile): of the
            // tuned kernel's builds only the ten-action one has that form
            p->last_dispatch = "k_search<false> (shape-generic; MSE head outside the ten-action build)";
            hipLaunchKernelGGL(k_search<false>, grid, block, s.lds_bytes, p->stream, s);
        } else
#ifdef MZ_DEV_SHAPES
        if (p->fast_planes == 512 && two) MZ_FAST(512, 2);
        else if (p->fast_planes == 256 && !two) MZ_FAST(256, 1);
        else hipLaunchKernelGGL(k_search<false>, grid, block, s.lds_bytes, p->stream, s);
#else
        if (p->fast_planes == 512) { if (two) MZ_FAST(512, 2); else MZ_FAST(512, 1); }
        else { if (two) MZ_FAST(256, 2); else MZ_FAST(256, 1); }
#endif
#undef MZ_FAST4
#undef MZ_FAST
    }
    else {
        p->last_dispatch = p->force_generic ? "k_search<false> (shape-generic, forced by MZ_FORCE_GENERIC=1)" : "k_search<false> (shape-generic: no tuned build for this shape)";
        hipLaunchKernelGGL(k_search<false>, grid, block, s.lds_bytes, p->stream, s);
    }
    HIPCHK(hipGetLastError());
    if (p->profiling) HIPCHK(hipEventRecord(eb, p->stream));
    return MZ_OK;
}

static int upload_roots(mz_planner* p, int batch, const float* h_obs, const uint8_t* h_mask, const int32_t* h_cur, const int32_t* h_opp,
                        const double* h_temp, const mz_rng_inputs* rng, int deterministic) {
    const mz_config& c = p->cfg;
    const size_t B = (size_t)batch, A = (size_t)c.num_actions;
    if (h_obs) HIPCHK(hipMemcpyAsync(p->d_obs, h_obs, B * obs_dim(c) * sizeof(float), hipMemcpyHostToDevice, p->stream));
    if (h_mask) HIPCHK(hipMemcpyAsync(p->d_mask, h_mask, B * A, hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipMemcpyAsync(p->d_cur, h_cur, B * sizeof(int), hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipMemcpyAsync(p->d_opp, h_opp, B * sizeof(int), hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipMemcpyAsync(p->d_temp, h_temp, B * sizeof(double), hipMemcpyHostToDevice, p->stream));
    if (rng) {
        const bool want_noise = !deterministic && c.root_dirichlet_alpha > 0.0 && c.root_exploration_eps > 0.0;
        if (want_noise) {
            if (!rng->h_noise) return fail(MZ_E_INVALID, "rng inputs given but h_noise is NULL while the search mixes Dirichlet noise");
            HIPCHK(hipMemcpyAsync(p->d_noise, rng->h_noise, B * A * sizeof(double), hipMemcpyHostToDevice, p->stream));
        }
        if (!rng->h_u_tie || (!deterministic && !rng->h_u_final)) return fail(MZ_E_INVALID, "rng inputs need h_u_tie and h_u_final");
        HIPCHK(hipMemcpyAsync(p->d_utie, rng->h_u_tie, B * (size_t)c.max_ties * sizeof(double), hipMemcpyHostToDevice, p->stream));
        if (rng->h_u_final) HIPCHK(hipMemcpyAsync(p->d_ufinal, rng->h_u_final, B * sizeof(double), hipMemcpyHostToDevice, p->stream));
    }
    return MZ_OK;
}

static int download_results(mz_planner* p, int batch, int32_t* h_action, double* h_pi, double* h_root, int32_t* h_visits) {
    const size_t B = (size_t)batch, A = (size_t)p->cfg.num_actions;
    int err = 0;
    if (h_action) HIPCHK(hipMemcpyAsync(h_action, p->d_action, B * sizeof(int), hipMemcpyDeviceToHost, p->stream));
    if (h_pi) HIPCHK(hipMemcpyAsync(h_pi, p->d_pi, B * A * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    if (h_root) HIPCHK(hipMemcpyAsync(h_root, p->d_root, B * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    if (h_visits) HIPCHK(hipMemcpyAsync(h_visits, p->d_visits, B * A * sizeof(int), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipMemcpyAsync(&err, p->d_err, sizeof(int), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    if (err) {
        HIPCHK(hipMemsetAsync(p->d_err, 0, sizeof(int), p->stream));
        HIPCHK(hipStreamSynchronize(p->stream));
        if (err == 4) return fail(MZ_E_TIES, "injected tie-break stream exhausted (raise mz_config.max_ties)");
        return fail(MZ_E_INVALID, "search kernel reported error " + std::to_string(err));
    }
    return MZ_OK;
}

extern "C" int mz_planner_search(mz_planner* p, int32_t batch, const float* h_obs, const uint8_t* h_mask, const int32_t* h_cur,
                                 const int32_t* h_opp, const double* h_temp, int32_t deterministic, const mz_rng_inputs* rng,
                                 int32_t* h_action, double* h_pi, double* h_root, int32_t* h_visits) {
    if (!p || !h_obs || !h_cur || !h_opp || !h_temp || !h_action || !h_pi || !h_root) return fail(MZ_E_INVALID, "null argument to mz_planner_search");
    if (batch < 1 || batch > p->cfg.num_envs) return fail(MZ_E_INVALID, "batch exceeds mz_config.num_envs");
    if (!p->committed) return fail(MZ_E_STATE, "weights not committed");
    HIPCHK(hipSetDevice(p->device));
    int rc = upload_roots(p, batch, h_obs, h_mask, h_cur, h_opp, h_temp, rng, deterministic);
    if (rc) return rc;
    rc = launch_search(p, batch, deterministic, h_mask != nullptr, rng != nullptr, false);
    if (rc) return rc;
    return download_results(p, batch, h_action, h_pi, h_root, h_visits);
}

extern "C" int mz_planner_search_scripted(mz_planner* p, int32_t batch, const float* h_pi0, const float* h_values, const float* h_rewards,
                                          const uint8_t* h_mask, const int32_t* h_cur, const int32_t* h_opp, const double* h_temp,
                                          int32_t deterministic, const mz_rng_inputs* rng, int32_t* h_action, double* h_pi, double* h_root,
                                          int32_t* h_visits, int32_t* h_tparent, int32_t* h_taction) {
    if (!p || !h_pi0 || !h_values || !h_rewards || !h_cur || !h_opp || !h_temp || !h_action || !h_pi || !h_root)
        return fail(MZ_E_INVALID, "null argument to mz_planner_search_scripted");
    if (batch < 1 || batch > p->cfg.num_envs) return fail(MZ_E_INVALID, "batch exceeds mz_config.num_envs");
    HIPCHK(hipSetDevice(p->device));
    const size_t B = (size_t)p->cfg.num_envs, A = (size_t)p->cfg.num_actions, S = (size_t)p->cfg.num_simulations;
    if (!p->d_spi0) {
        HIPCHK(hipMalloc(&p->d_spi0, B * A * sizeof(float)));
        HIPCHK(hipMalloc(&p->d_svalues, B * S * sizeof(float)));
        HIPCHK(hipMalloc(&p->d_srewards, B * S * sizeof(float)));
        HIPCHK(hipMalloc(&p->d_tparent, B * S * sizeof(int)));
        HIPCHK(hipMalloc(&p->d_taction, B * S * sizeof(int)));
    }
    const size_t b = (size_t)batch;
    HIPCHK(hipMemcpyAsync(p->d_spi0, h_pi0, b * A * sizeof(float), hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipMemcpyAsync(p->d_svalues, h_values, b * S * sizeof(float), hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipMemcpyAsync(p->d_srewards, h_rewards, b * S * sizeof(float), hipMemcpyHostToDevice, p->stream));
    int rc = upload_roots(p, batch, nullptr, h_mask, h_cur, h_opp, h_temp, rng, deterministic);
    if (rc) return rc;
    rc = launch_search(p, batch, deterministic, h_mask != nullptr, rng != nullptr, true);
    if (rc) return rc;
    if (h_tparent) HIPCHK(hipMemcpyAsync(h_tparent, p->d_tparent, b * S * sizeof(int), hipMemcpyDeviceToHost, p->stream));
    if (h_taction) HIPCHK(hipMemcpyAsync(h_taction, p->d_taction, b * S * sizeof(int), hipMemcpyDeviceToHost, p->stream));
    return download_results(p, batch, h_action, h_pi, h_root, h_visits);
}

// ---------------------------------------------------------------------------------------------------------
// device-resident self-play
// ---------------------------------------------------------------------------------------------------------
extern "C" int mz_selfplay_reset(mz_planner* p, int32_t env_kind, const double* h_init_state) {
    if (!p) return fail(MZ_E_INVALID, "null planner");
    HIPCHK(hipSetDevice(p->device));
    const mz_config& c = p->cfg;
    int board_n = 3, num_to_win = 3;
    if (env_kind == MZ_ENV_CARTPOLE) {
        if (c.num_actions != 2 || obs_dim(c) != 20) return fail(MZ_E_INVALID, "CartPole env needs num_actions == 2 and a (4,5) observation");
    } else if (env_kind == MZ_ENV_TICTACTOE) {
        if (c.num_actions != 10 || obs_dim(c) != 81) return fail(MZ_E_INVALID, "TicTacToe env needs num_actions == 10 and a (9,3,3) observation");
    } else if (env_kind == MZ_ENV_GOMOKU) {
        if (c.net_kind != MZ_NET_BOARD || c.obs_c != 9 || c.obs_h != c.obs_w || c.num_actions != c.obs_h * c.obs_w + 1 || c.obs_h < 5)
            return fail(MZ_E_INVALID, "Gomoku env needs a board net with a (9,N,N) observation, N >= 5, and num_actions == N*N + 1");
        board_n = c.obs_h;
        num_to_win = 5;
    } else if (env_kind == MZ_ENV_SYNTHETIC) {
        if (h_init_state) return fail(MZ_E_INVALID, "the synthetic env takes no initial state");
    } else {
        return fail(MZ_E_INVALID, "unknown env kind");
    }
    p->env_kind = env_kind;
    p->ring_len = (size_t)c.num_envs * obs_dim(c) * sizeof(float) * 64 > ((size_t)4 << 30) ? 16 : 64;  // record ring: at most a few GB
    if (p->has_replay) {
        // the record ring is every env's open trajectory: a whole board game, or the acc + unroll + td window (pipeline.py:118-121)
        // -- but never longer than an episode can get: the classic configs set acc_seq_length = 9999 ("never flush mid-episode", config.py:198), and
        // gym's TimeLimit ends CartPole after 500 steps (the synthetic frames env after 1000): 10 014 slots x 4096 envs of records would be ~5 GB
        const int limit = env_kind == MZ_ENV_CARTPOLE ? 500 : (env_kind == MZ_ENV_SYNTHETIC ? 1000 : (1 << 30) - 64);
        const int window = p->replay.acc + p->replay.K + p->replay.td, capped = limit + p->replay.K + p->replay.td;
        const int need = c.is_board_game ? c.num_actions + 1 : (window < capped ? window : capped);
        if (need > p->ring_len) p->ring_len = (need + 7) & ~7;
    }
    p->selfplay_moves = 0;
    p->ring_pos = 0;
    p->ring_count = 0;
    hipError_t e = env_alloc(p->env, env_kind, c.num_envs, c.num_actions, obs_dim(c), p->ring_len, board_n, num_to_win);
    if (e != hipSuccess) return fail(MZ_E_HIP, std::string("env_alloc: ") + hipGetErrorString(e));
    if (h_init_state) HIPCHK(hipMemcpyAsync(p->env.init_state, h_init_state, (size_t)c.num_envs * 4 * sizeof(double), hipMemcpyHostToDevice, p->stream));
    EnvLaunch L{};
    L.env = p->env; L.B = c.num_envs; L.seed = c.seed; L.use_init = h_init_state != nullptr;
    L.obs = p->d_obs; L.mask = p->d_mask; L.cur = p->d_cur; L.opp = p->d_opp;
    hipLaunchKernelGGL(k_env_reset, dim3((c.num_envs + 255) / 256), dim3(256), 0, p->stream, L);
    if (env_kind == MZ_ENV_SYNTHETIC)
        hipLaunchKernelGGL(k_env_synth_obs, dim3(((size_t)c.num_envs * ((obs_dim(c) + 3) / 4) + 255) / 256), dim3(256), 0, p->stream, L);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(p->stream));
    return MZ_OK;
}

extern "C" int mz_selfplay_step(mz_planner* p, double temperature, int32_t n_moves) {
    if (!p || n_moves < 1) return fail(MZ_E_INVALID, "bad argument to mz_selfplay_step");
    if (p->env_kind == MZ_ENV_NONE) return fail(MZ_E_STATE, "call mz_selfplay_reset first");
    if (!p->committed) return fail(MZ_E_STATE, "weights not committed");
    HIPCHK(hipSetDevice(p->device));
    const mz_config& c = p->cfg;
    auto epilogue = [&]() {
        if (p->has_replay) {
            EpiLaunch E{};
            E.env = p->env; E.ring = p->replay; E.B = c.num_envs; E.move_abs = p->selfplay_moves;
            hipLaunchKernelGGL(k_epi_scan, dim3(1), dim3(1024), 0, p->stream, E);
            hipLaunchKernelGGL(k_epilogue, dim3(c.num_envs), dim3(64), (size_t)p->ring_len * sizeof(double), p->stream, E);
            hipLaunchKernelGGL(k_epi_publish, dim3(1), dim3(1), 0, p->stream, p->replay);
        }
        p->selfplay_moves++;
    };
    for (int m = 0; m < n_moves; m++) {
        EnvLaunch L{};
        L.env = p->env; L.B = c.num_envs; L.seed = c.seed; L.temperature = temperature; L.move_counter = p->move_counter;
        L.obs = p->d_obs; L.mask = p->d_mask; L.cur = p->d_cur; L.opp = p->d_opp; L.temp_out = p->d_temp;
        L.action = p->d_action; L.pi = p->d_pi; L.root = p->d_root; L.slot = p->ring_pos; L.sims = c.num_simulations;
        if (!p->conv && !p->hbm_tree && p->fuse_env && p->env_kind != MZ_ENV_SYNTHETIC) {  // (synthetic frames are redrawn by k_env_synth_obs below)
            // MLP nets: the whole move -- temperature / record, search, env.step, auto-reset -- is ONE kernel launch
            int rc = launch_search(p, c.num_envs, 0, true, false, false, &L);
            if (rc) return rc;
            epilogue();
            p->ring_pos = (p->ring_pos + 1) % p->ring_len;
            if (p->ring_count < p->ring_len) p->ring_count++;
            continue;
        }
        // temperatures for this move, then the search, then env.step + record + auto-reset
        hipLaunchKernelGGL(k_env_pre, dim3((c.num_envs + 255) / 256), dim3(256), 0, p->stream, L);
        HIPCHK(hipMemcpyAsync(p->env.r_obs + (size_t)p->ring_pos * c.num_envs * obs_dim(c), p->d_obs, (size_t)c.num_envs * obs_dim(c) * sizeof(float),
                              hipMemcpyDeviceToDevice, p->stream));
        int rc = launch_search(p, c.num_envs, 0, true, false, false);
        if (rc) return rc;
        hipLaunchKernelGGL(k_env_step, dim3((c.num_envs + 15) / 16), dim3(256), 0, p->stream, L);
        if (p->env_kind == MZ_ENV_SYNTHETIC)
            hipLaunchKernelGGL(k_env_synth_obs, dim3(((size_t)c.num_envs * ((obs_dim(c) + 3) / 4) + 255) / 256), dim3(256), 0, p->stream, L);
        epilogue();
        HIPCHK(hipGetLastError());
        p->ring_pos = (p->ring_pos + 1) % p->ring_len;
        if (p->ring_count < p->ring_len) p->ring_count++;
    }
    return MZ_OK;
}

extern "C" int mz_selfplay_attach_replay(mz_planner* p, const mz_replay_ring* ring) {
    if (!p) return fail(MZ_E_INVALID, "null planner");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));  // attach / detach drain the planner: after a detach the caller owns counter and priorities again
    if (!ring) {
        p->has_replay = false;
        return MZ_OK;
    }
    const mz_config& c = p->cfg;
    if (ring->capacity < 1 || !ring->state || !ring->action || !ring->pi_prob || !ring->value || !ring->reward || !ring->priority || !ring->num_added)
        return fail(MZ_E_INVALID, "mz_replay_ring: capacity and every array but `origin` are required");
    if (ring->unroll_steps < 1 || ring->td_steps < 0 || ring->td_steps > 32 || ring->acc_seq_length < 1)
        return fail(MZ_E_INVALID, "mz_replay_ring: unroll_steps >= 1, 0 <= td_steps <= 32, acc_seq_length >= 1");
    {   // the epilogue kernels write through these pointers: every one must be memory of the planner's GPU (a host-resident replay would fault)
        const void* ptrs[8] = {ring->state, ring->action, ring->pi_prob, ring->value, ring->reward, ring->priority, ring->num_added, ring->origin};
        static const char* names[8] = {"state", "action", "pi_prob", "value", "reward", "priority", "num_added", "origin"};
        for (int i = 0; i < 8; i++) {
            if (!ptrs[i]) continue;  // (origin is optional)
            hipPointerAttribute_t at{};
            const hipError_t e = hipPointerGetAttributes(&at, ptrs[i]);
            if (e != hipSuccess || at.type != hipMemoryTypeDevice || at.device != p->device) {
                (void)hipGetLastError();
                return fail(MZ_E_INVALID, std::string("mz_replay_ring.") + names[i] + " is not memory of the planner's GPU");
            }
        }
    }
    ReplayRing& R = p->replay;
    R.capacity = ring->capacity; R.state = ring->state; R.action = reinterpret_cast<signed char*>(ring->action); R.action16 = c.num_actions > 128 ? 1 : 0; R.pi_prob = ring->pi_prob;
    R.value = ring->value; R.reward = ring->reward; R.priority = ring->priority; R.num_added = reinterpret_cast<long long*>(ring->num_added);
    R.origin = ring->origin; R.acc = ring->acc_seq_length; R.K = ring->unroll_steps; R.td = ring->td_steps; R.board = c.is_board_game;
    for (int i = 0; i <= R.td; i++) R.pw[i] = std::pow(c.discount, (double)i);  // Python's discount ** i (pipeline.py:663-666)
    // the device-owned write cursor starts at the caller's count; the caller's counter is from now on only PUBLISHED to
    if (!p->d_epi_ctr) HIPCHK(hipMalloc(&p->d_epi_ctr, 2 * sizeof(long long)));
    HIPCHK(hipMemsetAsync(p->d_epi_ctr, 0, 2 * sizeof(long long), p->stream));
    HIPCHK(hipMemcpyAsync(p->d_epi_ctr, R.num_added, sizeof(long long), hipMemcpyDeviceToDevice, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    R.ctr = p->d_epi_ctr;
    if (!p->d_epi_off) HIPCHK(hipMalloc(&p->d_epi_off, (size_t)c.num_envs * sizeof(int)));
    R.off = p->d_epi_off;
    p->has_replay = true;
    p->env_kind = MZ_ENV_NONE;  // the record ring must be re-sized: mz_selfplay_reset next
    return MZ_OK;
}

extern "C" int mz_selfplay_read(mz_planner* p, int32_t n_moves, float* h_obs, int32_t* h_action, float* h_reward, double* h_pi,
                                double* h_root, int32_t* h_player, uint8_t* h_done) {
    if (!p || n_moves < 1 || n_moves > p->ring_count) return fail(MZ_E_INVALID, "n_moves exceeds the recorded history");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    const size_t B = (size_t)p->cfg.num_envs, A = (size_t)p->cfg.num_actions, D = (size_t)obs_dim(p->cfg);
    // the ring is slot-major, so the last n_moves records are at most two contiguous slot ranges per field: 7 (or 14)
    // copies per call, not 7 per move
    const int first = ((p->ring_pos - n_moves) % p->ring_len + p->ring_len) % p->ring_len;
    const int n1 = n_moves < p->ring_len - first ? n_moves : p->ring_len - first;  // moves before the ring wraps
    for (int part = 0; part < 2; part++) {
        const size_t slot = part == 0 ? (size_t)first : 0, cnt = part == 0 ? (size_t)n1 : (size_t)(n_moves - n1), done_moves = part == 0 ? 0 : (size_t)n1;
        if (cnt == 0) continue;
        if (h_obs) HIPCHK(hipMemcpyAsync(h_obs + done_moves * B * D, p->env.r_obs + slot * B * D, cnt * B * D * sizeof(float), hipMemcpyDeviceToHost, p->stream));
        if (h_action) HIPCHK(hipMemcpyAsync(h_action + done_moves * B, p->env.r_action + slot * B, cnt * B * sizeof(int), hipMemcpyDeviceToHost, p->stream));
        if (h_reward) HIPCHK(hipMemcpyAsync(h_reward + done_moves * B, p->env.r_reward + slot * B, cnt * B * sizeof(float), hipMemcpyDeviceToHost, p->stream));
        if (h_pi) HIPCHK(hipMemcpyAsync(h_pi + done_moves * B * A, p->env.r_pi + slot * B * A, cnt * B * A * sizeof(double), hipMemcpyDeviceToHost, p->stream));
        if (h_root) HIPCHK(hipMemcpyAsync(h_root + done_moves * B, p->env.r_root + slot * B, cnt * B * sizeof(double), hipMemcpyDeviceToHost, p->stream));
        if (h_player) HIPCHK(hipMemcpyAsync(h_player + done_moves * B, p->env.r_player + slot * B, cnt * B * sizeof(int), hipMemcpyDeviceToHost, p->stream));
        if (h_done) HIPCHK(hipMemcpyAsync(h_done + done_moves * B, p->env.r_done + slot * B, cnt * B, hipMemcpyDeviceToHost, p->stream));
    }
    HIPCHK(hipStreamSynchronize(p->stream));
    return MZ_OK;
}

extern "C" int mz_selfplay_counters(mz_planner* p, int64_t out[4]) {
    if (!p || !out) return fail(MZ_E_INVALID, "null argument");
    if (p->env_kind == MZ_ENV_NONE) return fail(MZ_E_STATE, "call mz_selfplay_reset first");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    unsigned long long c[4];
    HIPCHK(hipMemcpy(c, p->env.counters, sizeof(c), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; i++) out[i] = (int64_t)c[i];
    return MZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// measurement hooks
// ---------------------------------------------------------------------------------------------------------
// diagnostic builds only (-DMZ_STAMPS): per-phase cycle sums of the last search launch, block 0.  Not part of the ABI header.
extern "C" int mz_debug_read_stamps(mz_planner* p, long long out[16]) {
    if (!p || !out) return fail(MZ_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipMemcpy(out, p->d_stamps, 16 * sizeof(long long), hipMemcpyDeviceToHost));
#ifdef MZ_STAMPS
    unsigned long long dbg[8];
    HIPCHK(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(mz::g_dbg), sizeof(dbg)));
    long long sub[8];
    HIPCHK(hipMemcpyFromSymbol(sub, HIP_SYMBOL(mz::g_sub), sizeof(sub)));
    for (int i = 0; i < 4; i++) out[11 + i] = sub[i];
    (void)dbg;
#endif
    return MZ_OK;
}

// diagnostic builds only (-DMZ_STAMPS): register-accumulated segment stamps of tree2_select [0..11] and tree2_backup [12..23]
// diagnostic (stamps build): every workgroup's duration [0..1023] and start stamp [1024..2047] of the last search launch
extern "C" int mz_debug_read_wg_cycles(mz_planner* p, long long out[2048]) {
    if (!p || !out) return fail(MZ_E_INVALID, "null argument");
    for (int i = 0; i < 2048; i++) out[i] = 0;
#ifdef MZ_STAMPS
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(mz::g_wg_cyc), 2048 * sizeof(long long)));
#endif
    return MZ_OK;
}

extern "C" int mz_debug_read_tree_stamps(mz_planner* p, long long out[32]) {
    if (!p || !out) return fail(MZ_E_INVALID, "null argument");
    for (int i = 0; i < 32; i++) out[i] = 0;
#ifdef MZ_STAMPS
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(mz::g_ts), 24 * sizeof(long long)));
    HIPCHK(hipMemcpyFromSymbol(out + 24, HIP_SYMBOL(mz::g_root_ts), 8 * sizeof(long long)));  // root inference segments (mz_mlp.h)
#endif
    return MZ_OK;
}

// test hooks, not part of the ABI header: capture the randomness a production-mode (on-device Philox) search consumes --
// normalised root Dirichlet noise, tie-break uniforms in the order they were drawn, the final action-sampling uniform -- in
// the layout of mz_rng_inputs, so a test can check their distributions and replay the same search in parity mode.
extern "C" int mz_debug_capture_rng(mz_planner* p, int32_t enable) {
    if (!p) return fail(MZ_E_INVALID, "null planner");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    const size_t B = (size_t)p->cfg.num_envs, A = (size_t)p->cfg.num_actions, T = (size_t)p->cfg.max_ties;
    if (enable && !p->d_dbg_noise) {
        HIPCHK(hipMalloc(&p->d_dbg_noise, B * A * sizeof(double)));
        HIPCHK(hipMalloc(&p->d_dbg_utie, B * T * sizeof(double)));
        HIPCHK(hipMalloc(&p->d_dbg_ufinal, B * sizeof(double)));
    }
    if (!enable) {
        if (p->d_dbg_noise) { (void)hipFree(p->d_dbg_noise); (void)hipFree(p->d_dbg_utie); (void)hipFree(p->d_dbg_ufinal); }
        p->d_dbg_noise = p->d_dbg_utie = p->d_dbg_ufinal = nullptr;
    } else {  // 0.5 where a search draws nothing: what the parity-mode tests inject for unused slots
        std::vector<double> half(B * (T > A ? T : A), 0.5);
        HIPCHK(hipMemcpy(p->d_dbg_utie, half.data(), B * T * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(p->d_dbg_ufinal, half.data(), B * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemset(p->d_dbg_noise, 0, B * A * sizeof(double)));
        HIPCHK(hipDeviceSynchronize());
    }
    return MZ_OK;
}

extern "C" int mz_debug_read_rng(mz_planner* p, double* h_noise, double* h_utie, double* h_ufinal) {
    if (!p || !p->d_dbg_noise) return fail(MZ_E_STATE, "mz_debug_capture_rng(p, 1) first");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    const size_t B = (size_t)p->cfg.num_envs, A = (size_t)p->cfg.num_actions, T = (size_t)p->cfg.max_ties;
    if (h_noise) HIPCHK(hipMemcpy(h_noise, p->d_dbg_noise, B * A * sizeof(double), hipMemcpyDeviceToHost));
    if (h_utie) HIPCHK(hipMemcpy(h_utie, p->d_dbg_utie, B * T * sizeof(double), hipMemcpyDeviceToHost));
    if (h_ufinal) HIPCHK(hipMemcpy(h_ufinal, p->d_dbg_ufinal, B * sizeof(double), hipMemcpyDeviceToHost));
    return MZ_OK;
}

// test hook, not part of the ABI header: overwrite the per-env step counters of the running episodes (lets a test reach the
// TimeLimit-500 truncation of CartPole without a policy that balances the pole for 500 steps)
extern "C" int mz_debug_set_env_steps(mz_planner* p, const int32_t* h_steps) {
    if (!p || !h_steps) return fail(MZ_E_INVALID, "null argument");
    if (p->env_kind == MZ_ENV_NONE) return fail(MZ_E_STATE, "call mz_selfplay_reset first");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipMemcpy(p->env.steps, h_steps, (size_t)p->cfg.num_envs * sizeof(int), hipMemcpyHostToDevice));
    return MZ_OK;
}

// diagnostic builds only (-DMZ_STAMPS -DMZ_COUNTERS): tree counters [levels, cache hits, descents, min-max changes]
extern "C" int mz_debug_read_counters(mz_planner* p, long long out[8]) {
    if (!p || !out) return fail(MZ_E_INVALID, "null argument");
    for (int i = 0; i < 8; i++) out[i] = 0;
#ifdef MZ_STAMPS
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    unsigned long long dbg[8];
    HIPCHK(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(mz::g_dbg), sizeof(dbg)));
    for (int i = 0; i < 8; i++) out[i] = (long long)dbg[i];
    long long sub[8];
    HIPCHK(hipMemcpyFromSymbol(sub, HIP_SYMBOL(mz::g_sub), sizeof(sub)));
    (void)sub;
#endif
    return MZ_OK;
}

extern "C" int mz_planner_synchronize(mz_planner* p) {
    if (!p) return fail(MZ_E_INVALID, "null planner");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    return MZ_OK;
}

extern "C" int mz_profile_begin(mz_planner* p) {
    if (!p) return fail(MZ_E_INVALID, "null planner");
    HIPCHK(hipSetDevice(p->device));
    p->profiling = true;
    p->kev_used = 0;
    HIPCHK(hipEventRecord(p->ev_begin, p->stream));
    return MZ_OK;
}

extern "C" int mz_profile_end(mz_planner* p, double* elapsed_ms, double* search_kernel_ms, int64_t* search_kernel_launches) {
    if (!p || !p->profiling) return fail(MZ_E_STATE, "mz_profile_end without mz_profile_begin");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipEventRecord(p->ev_end, p->stream));
    HIPCHK(hipEventSynchronize(p->ev_end));
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, p->ev_begin, p->ev_end));
    if (elapsed_ms) *elapsed_ms = ms;
    double ksum = 0.0;
    for (size_t i = 0; i < p->kev_used; i++) {
        float k = 0.0f;
        HIPCHK(hipEventElapsedTime(&k, p->kev[i].first, p->kev[i].second));
        ksum += k;
    }
    if (search_kernel_ms) *search_kernel_ms = ksum;
    if (search_kernel_launches) *search_kernel_launches = (int64_t)p->kev_used;
    p->profiling = false;
    return MZ_OK;
}
