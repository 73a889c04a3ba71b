// mz_search.h -- the fused, persistent-per-move search kernel for MLP-class configs (CartPole / LunarLander /
// TicTacToe shapes): one 256-thread workgroup owns 16 environments and runs ALL num_simulations of their searches
// (mcts.py:352-407) without leaving the CU:
//   * the 16 trees (node statistics, child tables, root priors, min-max stats) live in LDS as structure-of-arrays;
//   * selection (mcts.py:104-127,159-200) uses 16 lanes per environment (lanes over actions), wavefront-segment
//     max/ballot reductions for the pUCT argmax and its tie set;
//   * the leaf evaluation (network.py:86-111) is the MFMA tile pipeline of mz_mlp.h, weights streamed from L2;
//   * expansion + backup (mcts.py:129-157,386-389) walk the LDS tree in float64 exactly like the reference's
//     Python floats; hidden states go to an HBM node store hidden[env][node][H] (L2/MALL resident).
// Environments are independent (mcts.py has no cross-env state), so no inter-workgroup communication exists.
#pragma once
#include "mz_env.h"
#include "mz_mlp.h"

namespace mz {

struct __attribute__((aligned(8))) TreeNode {  // 32 bytes
    double W;       // sum of backed-up values          (Node.W, mcts.py:69)
    double vq;      // reward + discount * (+/-)Q, refreshed by backup; it is both the min-max update value
                    // (mcts.py:147-150) and the un-normalised child_Q term (mcts.py:174)
    int N;          // visit count                       (Node.N, mcts.py:68)
    float reward;   // float32 network output, widened on use (mcts.py:96)
    short parent;
    short move;
    int player;
};

struct SearchParams {
    MlpNet net;
    MlpLds o;
    // LDS byte offsets of the tree part
    int t_nodes, t_child, t_prior, t_pi0, t_mm, t_sel, t_ftab, t_ptr, t_tmp;
    // tree_mode 2 layout (mz_tree2.h): 16-byte nodes, per-(node, action) child entries, selection cache, path (t_ver: unused, kept for layout stability)
    int t2_nodes, t2_entries, t_cache, t_path, t_ver, t2_ftab;
    int tree_mode;  // 0: reference-order tree walk (mz_search.h); 2: entry table + selection cache + lane-parallel backup (mz_tree2.h, A <= 16)
    const double* ftab_tri;  // [(S+1)(S+2)/2]: ftab restricted to n_child <= N
    int lds_bytes;
    // search configuration (config.py:58-78)
    int S, A, NN;
    double discount;
    int board, has_bounds;
    double kb_min, kb_max, alpha, eps;
    int deterministic, has_mask;
    int noise_mode;  // 0: none, 1: injected, 2: Philox Dirichlet on device
    int legacy_promo;  // mz_config.legacy_scalar_promotion: child_U's product in float64 even where the prior stayed float32 (numpy 1.21)
    int rng_mode;    // 0: injected tie/final uniforms, 1: Philox
    int max_ties;
    // batch
    int B;
    const float* obs;       // [B][in_dim]
    const unsigned char* mask;  // [B][A]
    const int* cur;
    const int* opp;
    const double* temperature;
    const double* noise;    // [B][A]
    const double* u_tie;    // [B][max_ties]
    const double* u_final;  // [B]
    float* hidden;          // [B][NN][H] node store
    const double* ftab;     // [(S+1)][(S+1)]: ((log((N+base+1)/base)+c_init)*sqrt(N)) / (n_child+1), host-computed float64
    int* out_action;
    double* out_pi;
    double* out_root;
    int* out_visits;
    int* err;
    // scripted-network test hook
    const float* s_pi0;
    const float* s_values;
    const float* s_rewards;
    int* trace_parent;
    int* trace_action;
    unsigned long long seed;
    unsigned int move_counter;
    unsigned int env_offset;  // global id of env 0 (multi-GPU sharding: Philox streams are keyed by global env id)
    // test hook (mz_debug_capture_rng): when set, production-mode (Philox) searches store the draws they consume in the layout
    // of the injected-randomness inputs, so that a test can replay the very same search in parity mode
    double* dbg_noise;        // [B][A] normalised Dirichlet noise
    double* dbg_utie;         // [B][max_ties]
    double* dbg_ufinal;       // [B]
    long long* stamps;        // diagnostic builds (-DMZ_STAMPS) only: per-phase cycle sums of block 0, else unused
    // device self-play with the environment fused into the search kernel (one launch per lock-step move): the env's first lane
    // runs env_pre_one before the search (temperature, record of player / observation) and env_step_group after it
    int fuse_env;
    EnvLaunch fenv;
    // k_search_fast with helper waves (mz_search_fast.h): bit 0: the helpers normalise the new state, bit 1: they reduce the reward row
    int hwx;
};

// the pre-search half of a fused self-play move for env e (global id env_g); call with the env's 16 lanes
__device__ __forceinline__ void fused_env_pre(const SearchParams& P, int a0, int env_g, bool env_ok) {
    if (!P.fuse_env || !env_ok) return;
    const int D = P.fenv.env.D;
    float* ro = P.fenv.env.r_obs + ((size_t)P.fenv.slot * P.fenv.B + env_g) * D;
    const float* o = P.fenv.obs + (size_t)env_g * D;
    for (int i = a0; i < D; i += 16) ro[i] = o[i];
    if (a0 == 0) env_pre_one(P.fenv, env_g);
}

// Phase stamps for the diagnostic build (python -m muzero_amd.build --stamps -> libmzplanner_hip_stamps.so): thread 0
// of block 0 accumulates s_memtime deltas per phase.  Never compiled into the product library; read SHARES, not totals.
#ifdef MZ_STAMPS
#ifdef MZ_COUNTERS
#define MZ_COUNT(i, v) atomicAdd(&g_dbg[i], (unsigned long long)(v))
#else
#define MZ_COUNT(i, v) do {} while (0)
#endif
__device__ unsigned long long g_dbg[8];
__device__ long long g_sub[8];  // sub-phase cycle sums written by thread 0 of block 0 only
#define MZ_SUB_DECL long long _s0 = 0;
#define MZ_SUB_START() do { if (blockIdx.x == 0 && threadIdx.x == 0) _s0 = __builtin_readcyclecounter(); } while (0)
#define MZ_SUB(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { long long _s1 = __builtin_readcyclecounter(); g_sub[i] += _s1 - _s0; _s0 = _s1; } } while (0)  // [0] levels visited, [1] cache hits, [2] descents, [3] version bumps, [4] max-depth sum per wave-descent
// fine-grained segment stamps of the tree functions (block 0, wave 0): cycles and event counts accumulate in REGISTERS
// and reach memory once per call (a global read-modify-write per event costs more than the segments it would time)
__device__ long long g_ts[32];
__device__ __forceinline__ long long ts_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return (long long)t;
}
#define MZ_TS_DECL long long _ts0 = 0; long long _tsa[12] = {0}; const bool _tson = blockIdx.x == 0 && threadIdx.x < 64;
#define MZ_TS_START() do { if (_tson) _ts0 = ts_now(); } while (0)
#define MZ_TS(i) do { if (_tson) { const long long _n = ts_now(); _tsa[i] += _n - _ts0; _ts0 = _n; } } while (0)
#define MZ_TS_COUNT(i) do { if (_tson) _tsa[i] += 1; } while (0)
#define MZ_TS_FLUSH(base) do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int _i = 0; _i < 12; _i++) g_ts[(base) + _i] += _tsa[_i]; } while (0)
#define MZ_SUBX_START() do {} while (0)
#define MZ_SUBX(i) do {} while (0)
#define MZ_SUBX_COUNT(i) do {} while (0)
// (phase sums accumulate in global memory with return-less atomics, not in registers: 16 64-bit accumulators per thread pushed the
// 8-wave kernels -- 256 registers per wave -- into VGPR spills that distorted exactly the phases being timed)
__device__ long long g_stamp_acc[16];
// round 6: every workgroup's own duration (cycles from its first to its last instruction) and start offset, last launch: how far apart do the 256
// workgroups of a move finish?  (a kernel ends with its slowest workgroup)
__device__ long long g_wg_cyc[2048];
#define MZ_STAMP_DECL long long _t0 = 0; const long long _wg0 = (long long)__builtin_readcyclecounter();
#define MZ_STAMP_START() do { if (blockIdx.x == 0 && threadIdx.x == 0) _t0 = __builtin_readcyclecounter(); } while (0)
#define MZ_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { long long _t1 = __builtin_readcyclecounter(); atomicAdd(reinterpret_cast<unsigned long long*>(&g_stamp_acc[i]), (unsigned long long)(_t1 - _t0)); _t0 = _t1; } } while (0)
#define MZ_STAMP_FLUSH(P) do { if (threadIdx.x == 0 && blockIdx.x < 1024) { g_wg_cyc[blockIdx.x] = (long long)__builtin_readcyclecounter() - _wg0; g_wg_cyc[1024 + blockIdx.x] = _wg0; } \
    if (blockIdx.x == 0 && threadIdx.x == 0 && (P).stamps) { __threadfence(); for (int _i = 0; _i < 16; _i++) (P).stamps[_i] = (long long)atomicExch(reinterpret_cast<unsigned long long*>(&g_stamp_acc[_i]), 0ULL); } } while (0)
#else
#define MZ_COUNT(i, v) do {} while (0)
#define MZ_SUB_DECL
#define MZ_SUB_START() do {} while (0)
#define MZ_SUB(i) do {} while (0)
#define MZ_SUBX_START() do {} while (0)
#define MZ_SUBX(i) do {} while (0)
#define MZ_SUBX_COUNT(i) do {} while (0)
#define MZ_TS_DECL
#define MZ_TS_START() do {} while (0)
#define MZ_TS(i) do {} while (0)
#define MZ_TS_COUNT(i) do {} while (0)
#define MZ_TS_FLUSH(base) do {} while (0)
#define MZ_STAMP_DECL
#define MZ_STAMP_START() do {} while (0)
#define MZ_STAMP(i) do {} while (0)
#define MZ_STAMP_FLUSH(P) do {} while (0)
#endif

__device__ __forceinline__ TreeNode* node_at(unsigned char* smem, const SearchParams& P, int e, int i) {
    return reinterpret_cast<TreeNode*>(smem + P.t_nodes) + (e * P.NN + i);
}
__device__ __forceinline__ short* child_row(unsigned char* smem, const SearchParams& P, int e, int i) {
    return reinterpret_cast<short*>(smem + P.t_child) + (size_t)(e * P.NN + i) * P.A;
}

__device__ __forceinline__ int nth_set_bit(unsigned m, int idx) {
    for (int i = 0; i < idx; i++) m &= m - 1;
    return __ffs(m) - 1;
}

constexpr int MAX_CH = 4;  // action chunks of 16 lanes: A <= 64 in the LDS-resident kernel

// pUCT evaluation of ONE node for the env's 16-lane segment (best_child, mcts.py:104-127): child_Q (mcts.py:159-178) +
// child_U (mcts.py:180-200) per action lane, segment max, tie set in ascending action order, np.random.choice among
// real ties.  `draw` says whether this segment may consume a tie-break draw.  Returns the selected action
// (segment-uniform).  Must be executed by all 64 lanes of the wave (DPP / ballot inside).
template <int MAXCH = MAX_CH>
__device__ __forceinline__ int select_level(unsigned char* smem, const SearchParams& P, int e, int a0, int seg, int n, double mn, double mx,
                                            bool draw, int& ties, int env_g) {
    const double* ftab = reinterpret_cast<const double*>(smem + P.t_ftab);
    const double* prior = reinterpret_cast<const double*>(smem + P.t_prior) + e * P.A;
    const bool norm = mx > mn;
    const bool prior_f32 = (P.noise_mode == 0 && !P.legacy_promo);
    const int nch = (P.A + 15) >> 4;
    const int Nn = node_at(smem, P, e, n)->N;
    const double* frow = ftab + Nn * (P.S + 1);
    const short* crow = child_row(smem, P, e, n);
    float u[MAXCH];
    float best = __uint_as_float(0xff800000u);
#pragma unroll
    for (int ch = 0; ch < MAXCH; ch++) {
        u[ch] = __uint_as_float(0xff800000u);
        const int a = ch * 16 + a0;
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
    best = butterfly16_max(best);  // DPP row rotations inside the env's 16-lane segment
    // tie set in ascending action order (np.where(ucb == max), mcts.py:124)
    unsigned msk[MAXCH];
    int total = 0;
#pragma unroll
    for (int ch = 0; ch < MAXCH; ch++) {
        const int a = ch * 16 + a0;
        const bool eq = (ch < nch) && (a < P.A) && (u[ch] == best);
        const unsigned long long bal = __ballot(eq);
        msk[ch] = (unsigned)(bal >> (16 * seg)) & 0xffffu;
        total += __popc(msk[ch]);
    }
    int pick = 0;
    if (draw && total > 1) {  // np.random.choice consumes randomness only when there is a real tie
        double uu;
        if (P.rng_mode == 0) {
            if (ties < P.max_ties) uu = P.u_tie[(size_t)env_g * P.max_ties + ties];
            else { uu = 0.5; if (a0 == 0) atomicExch(P.err, 4); }
        } else {
            Philox g(P.seed, P.env_offset + (unsigned)env_g, P.move_counter, 0x10000000u + (unsigned)ties);
            uu = g.uniform();
            if (P.dbg_utie && a0 == 0 && ties < P.max_ties) P.dbg_utie[(size_t)env_g * P.max_ties + ties] = uu;
        }
        ties++;
        pick = (int)floor(uu * (double)total);
        pick = pick >= total ? total - 1 : pick;
    }
    int a_sel = 0, cum = 0;
    bool found = false;
#pragma unroll
    for (int ch = 0; ch < MAXCH; ch++) {
        const int c = __popc(msk[ch]);
        if (!found && pick < cum + c) {
            a_sel = ch * 16 + nth_set_bit(msk[ch], pick - cum);
            found = true;
        }
        cum += c;
    }
    return a_sel;
}

// One descent from the root to an unexpanded child for all 16 envs of the tile (mcts.py:372-379).
// Every lane of an env's 16-lane segment ends with identical (segment-uniform) results.
template <int MAXCH = MAX_CH>
__device__ __forceinline__ void tree_select(unsigned char* smem, const SearchParams& P, int tid, bool env_ok, int env_g, int& leaf_parent,
                                            int& leaf_action) {
    const int e = tid >> 4, a0 = tid & 15, seg = (tid & 63) >> 4;
    double* mm = reinterpret_cast<double*>(smem + P.t_mm) + e * 2;
    int* sel = reinterpret_cast<int*>(smem + P.t_sel) + e * 4;
    const double mn = mm[0], mx = mm[1];
    int n = 0, cp = env_ok ? P.cur[env_g] : 0, op = env_ok ? P.opp[env_g] : 0;
    int ties = sel[3];
    bool done = !env_ok;
    int lp = 0, la = 0, lpl = 0, depth = 0;
    while (__any(!done)) {
        const int a_sel = select_level<MAXCH>(smem, P, e, a0, seg, n, mn, mx, !done, ties, env_g);
        const int t = cp; cp = op; op = t;  // mcts.py:379
        if (!done) {
            const int c = child_row(smem, P, e, n)[a_sel];
            depth++;
            if (c < 0 || depth > P.NN) {
                done = true;
                lp = n; la = a_sel; lpl = cp;
            } else {
                n = c;
            }
        }
    }
    if (a0 == 0) {
        sel[0] = lp; sel[1] = la; sel[2] = lpl; sel[3] = ties;
    }
    leaf_parent = lp;
    leaf_action = la;
}

// expand the selected leaf and back its value up to the root (mcts.py:386-389, 129-157); one lane per env
__device__ __forceinline__ void tree_expand_backup(unsigned char* smem, const SearchParams& P, int e, int s, float r32, float v32) {
    int* sel = reinterpret_cast<int*>(smem + P.t_sel) + e * 4;
    double* mm = reinterpret_cast<double*>(smem + P.t_mm) + e * 2;
    const int n = sel[0], a = sel[1], cp = sel[2], nw = s + 1;
    child_row(smem, P, e, n)[a] = (short)nw;
    TreeNode* nd = node_at(smem, P, e, nw);
    nd->W = 0.0; nd->vq = 0.0; nd->N = 0; nd->reward = r32; nd->parent = (short)n; nd->move = (short)a; nd->player = cp;
    double val = (double)v32, mn = mm[0], mx = mm[1];
    const double g = P.discount;
    for (int c = nw; c >= 0;) {
        TreeNode* x = node_at(smem, P, e, c);
        const bool same = (x->player == cp);
        const double W = x->W + (same ? val : -val);
        const int N = x->N + 1;
        const double rw = (double)x->reward;
        const double Q = W / (double)N;
        const double v = P.board ? (rw + g * -Q) : (rw + g * Q);
        x->W = W; x->N = N; x->vq = v;
        mx = v > mx ? v : mx;
        mn = v < mn ? v : mn;
        val = (P.board && same) ? (-rw + g * val) : (rw + g * val);
        c = x->parent;
    }
    mm[0] = mn; mm[1] = mx;
}

// Production randomness (noise_mode 2): the gamma(alpha) draws of the root's Dirichlet noise (mcts.py:245), one action per
// lane of the env's 16-lane segment, each from its own Philox stream keyed by (seed, env, move, action) -- the rejection
// sampler is ~10 k cycles per draw when one lane does all A of them.  Call with every thread, then barrier, then root_prior.
__device__ __forceinline__ void root_noise_lanes(unsigned char* smem, const SearchParams& P, int e, int a0, int env_g, bool env_ok) {
    if (P.noise_mode != 2 || !env_ok) return;
    double* tmp = reinterpret_cast<double*>(smem + P.t_tmp) + e * P.A;
    for (int a = a0; a < P.A; a += 16) {
        Philox g(P.seed, P.env_offset + (unsigned)env_g, P.move_counter, 0x20000000u + (unsigned)a);
        tmp[a] = gamma_sample(g, P.alpha);
    }
}

// root prior: Dirichlet mix + illegal-action mask + renormalisation (mcts.py:357-365, 244-247, 293-299); one lane per env
__device__ __forceinline__ void root_prior(unsigned char* smem, const SearchParams& P, int e, int env_g) {
    double* prior = reinterpret_cast<double*>(smem + P.t_prior) + e * P.A;
    float* pi0 = reinterpret_cast<float*>(smem + P.t_pi0) + e * P.A;
    double* tmp = reinterpret_cast<double*>(smem + P.t_tmp) + e * P.A;
    const unsigned char* mk = P.has_mask ? P.mask + (size_t)env_g * P.A : nullptr;
    const int A = P.A;
    if (P.noise_mode != 0) {
        if (P.noise_mode == 1) {
            for (int a = 0; a < A; a++) tmp[a] = P.noise[(size_t)env_g * A + a];
        } else {  // tmp[] holds the gamma draws of root_noise_lanes
            double s = 0.0;
            for (int a = 0; a < A; a++) s += tmp[a];
            for (int a = 0; a < A; a++) tmp[a] = s > 0.0 ? tmp[a] / s : 1.0 / (double)A;
            if (P.dbg_noise)
                for (int a = 0; a < A; a++) P.dbg_noise[(size_t)env_g * A + a] = tmp[a];
        }
        const float om = (float)(1.0 - P.eps);
        for (int a = 0; a < A; a++) {
            const float t = om * pi0[a];        // float32 product (python scalar * float32 array)
            const double en = P.eps * tmp[a];   // float64
            prior[a] = (double)t + en;
        }
        if (mk) {
            for (int a = 0; a < A; a++)
                if (!mk[a]) prior[a] = 0.0;
            const double s = np_sum_f64(prior, A);
            if (s > 0)
                for (int a = 0; a < A; a++) prior[a] = prior[a] / s;
        }
    } else {
        if (mk) {
            for (int a = 0; a < A; a++)
                if (!mk[a]) pi0[a] = 0.0f;
            const float s = np_sum_f32(pi0, A);
            if (s > 0)
                for (int a = 0; a < A; a++) pi0[a] = pi0[a] / s;
        }
        for (int a = 0; a < A; a++) prior[a] = (double)pi0[a];
    }
}

// play: visit counts -> policy -> action (mcts.py:391-407); one lane per env
// play: visit counts -> policy -> action (mcts.py:391-407); one lane per env.  raw_visits: the root children's N in LDS.
__device__ __forceinline__ void play_from_visits(unsigned char* smem, const SearchParams& P, int e, int env_g, const int* raw_visits, double rootW,
                                                 int rootN) {
    double* tmp = reinterpret_cast<double*>(smem + P.t_tmp) + e * P.A;
    const unsigned char* mk = P.has_mask ? P.mask + (size_t)env_g * P.A : nullptr;
    const int A = P.A;
    const double T = P.temperature[env_g];
    double ex = 1.0;
    if (T > 0.0) {
        ex = 1.0 / T;
        ex = ex < 5.0 ? ex : 5.0;
        ex = ex > 1.0 ? ex : 1.0;
    }
    int best = 0, bestv = -1;
    for (int a = 0; a < A; a++) {
        int v = raw_visits[a];
        if (mk && !mk[a]) v = 0;
        if (P.out_visits) P.out_visits[(size_t)env_g * A + a] = v;
        if (v > bestv) { bestv = v; best = a; }
        tmp[a] = (T > 0.0) ? pow_policy((double)v, ex) : (double)v;
    }
    double s = np_sum_f64(tmp, A);
    if (!(s > 0.0)) {
        // Every visit went to an illegal root child: the first simulation of a search ties ALL actions (U == 0 while the
        // root has N == 0, mcts.py:193-195) and may pick an illegal one, which then keeps winning on Q alone when simulations
        // are few.  The reference divides 0/0 here (a NaN policy) and np.random.choice then raises ValueError
        // (mcts.py:279,404).  Parity mode (injected draws) returns the same NaN policy -- the Python mirror raises the
        // ValueError; production self-play plays uniformly over the legal actions instead of stopping the actor.
        if (P.rng_mode != 0) {
            s = 0.0;
            best = -1;
            for (int a = 0; a < A; a++) {
                const bool legal = !mk || mk[a];
                tmp[a] = legal ? 1.0 : 0.0;
                s = s + tmp[a];
                if (legal && best < 0) best = a;
            }
            best = best < 0 ? 0 : best;
        }
    }
    double* pi = P.out_pi + (size_t)env_g * A;
    for (int a = 0; a < A; a++) pi[a] = tmp[a] / s;
    int action = best;
    if (!P.deterministic) {
        double uu;
        if (P.rng_mode == 0) uu = P.u_final[env_g];
        else {
            Philox g(P.seed, P.env_offset + (unsigned)env_g, P.move_counter, 0x30000000u);
            uu = g.uniform();
            if (P.dbg_ufinal) P.dbg_ufinal[env_g] = uu;
        }
        // np.random.choice(p=pi): cdf = cumsum(pi); cdf /= cdf[-1]; searchsorted(cdf, u, side='right')
        double c = 0.0;
        for (int a = 0; a < A; a++) { c = c + pi[a]; tmp[a] = c; }
        const double last = tmp[A - 1];
        int idx = 0;
        for (int a = 0; a < A; a++)
            if (tmp[a] / last <= uu) idx = a + 1;
        action = idx >= A ? A - 1 : idx;
    }
    P.out_action[env_g] = action;
    P.out_root[env_g] = rootN > 0 ? rootW / (double)rootN : 0.0;
}

__device__ __forceinline__ void tree_finish(unsigned char* smem, const SearchParams& P, int e, int env_g) {
    int* rv = reinterpret_cast<int*>(smem + P.t_pi0) + e * P.A;  // the float32 root policy is no longer needed: reuse as int scratch
    const short* crow = child_row(smem, P, e, 0);
    for (int a = 0; a < P.A; a++) {
        const int c = crow[a];
        rv[a] = c >= 0 ? node_at(smem, P, e, c)->N : 0;
    }
    const TreeNode* root = node_at(smem, P, e, 0);
    play_from_visits(smem, P, e, env_g, rv, root->W, root->N);
}

#include "mz_tree2.h"

template <bool SCRIPTED>
__global__ __launch_bounds__(WG_THREADS) void k_search(const SearchParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, e = tid >> 4, a0 = tid & 15;
    const int env_g = blockIdx.x * TILE_E + e;
    const bool env_ok = env_g < P.B;
    if (!SCRIPTED) fused_env_pre(P, a0, env_g, env_ok);
    float* pi0 = reinterpret_cast<float*>(smem + P.t_pi0);
    const float** src = reinterpret_cast<const float**>(smem + P.t_ptr);
    float** dst = reinterpret_cast<float**>(smem + P.t_ptr) + 16;

    // tables and tree initialisation
    if (!SCRIPTED) stage_biases(P.net, lds, tid);
    {
        if (P.tree_mode == 2) {
            tree2_init(smem, P, tid, env_ok, env_g);
        } else {
            double* ft = reinterpret_cast<double*>(smem + P.t_ftab);
            for (int i = tid; i < (P.S + 1) * (P.S + 1); i += WG_THREADS) ft[i] = P.ftab[i];
            short* ch = reinterpret_cast<short*>(smem + P.t_child);
            for (int i = tid; i < TILE_E * P.NN * P.A; i += WG_THREADS) ch[i] = -1;
            if (a0 == 0) {
                TreeNode* r = node_at(smem, P, e, 0);
                r->W = 0.0; r->vq = 0.0; r->N = 0; r->reward = 0.0f; r->parent = -1; r->move = -1;
                r->player = env_ok ? P.cur[env_g] : 0;
            }
        }
        if (a0 == 0) {
            double* mm = reinterpret_cast<double*>(smem + P.t_mm) + e * 2;
            mm[0] = P.has_bounds ? P.kb_min : __longlong_as_double(0x7ff0000000000000LL);   // MinMaxStats, mcts.py:36-38
            mm[1] = P.has_bounds ? P.kb_max : __longlong_as_double(0xfff0000000000000LL);
            int* sel = reinterpret_cast<int*>(smem + P.t_sel) + e * 4;
            sel[0] = sel[1] = sel[2] = sel[3] = 0;
            if (!SCRIPTED) {
                src[e] = env_ok ? P.obs + (size_t)env_g * P.net.in_dim : nullptr;
                dst[e] = env_ok ? P.hidden + (size_t)env_g * P.NN * P.net.H : nullptr;
            }
        }
    }
    __syncthreads();
    if (SCRIPTED) {
        for (int i = tid; i < TILE_E * P.A; i += WG_THREADS) {
            const int ee = i / P.A, a = i - ee * P.A, eg = blockIdx.x * TILE_E + ee;
            pi0[i] = eg < P.B ? P.s_pi0[(size_t)eg * P.A + a] : 1.0f / (float)P.A;
        }
    } else {
        load_obs(P.net, lds + P.o.X, src, tid);
        __syncthreads();
        mlp_initial_tile(P.net, P.o, lds, dst, pi0, tid, true, false);  // root value is discarded (mcts.py:356-367)
    }
    root_noise_lanes(smem, P, e, a0, env_g, env_ok);
    __syncthreads();
    if (a0 == 0 && env_ok) root_prior(smem, P, e, env_g);
    __syncthreads();

    int resume = 0;
    const int cp0 = env_ok ? P.cur[env_g] : 0, op0 = env_ok ? P.opp[env_g] : 0;  // the root's players: read once per move, not per descent
    Tree2Env T;  // tree_mode 2: the env's search state lives in its lanes' registers
    tree2_env_init(T, P);
    for (int s = 0; s < P.S; s++) {
        int lp_unused, la_unused;
        if (P.tree_mode == 2) tree2_select<true>(smem, P, tid, env_ok, env_g, T, cp0, op0, resume);
        else tree_select(smem, P, tid, env_ok, env_g, lp_unused, la_unused);
        __syncthreads();
        const int* sel = reinterpret_cast<const int*>(smem + P.t_sel) + e * 4;
        float r32, v32;
        if (SCRIPTED) {
            r32 = env_ok ? P.s_rewards[(size_t)env_g * P.S + s] : 0.0f;
            v32 = env_ok ? P.s_values[(size_t)env_g * P.S + s] : 0.0f;
            if (a0 == 0 && env_ok) {
                P.trace_parent[(size_t)env_g * P.S + s] = sel[0];
                P.trace_action[(size_t)env_g * P.S + s] = sel[1];
            }
        } else {
            int* act = reinterpret_cast<int*>(smem + P.t_sel) + 64;  // [16] leaf actions
            if (a0 == 0) {
                float* base = env_ok ? P.hidden + (size_t)env_g * P.NN * P.net.H : nullptr;
                src[e] = env_ok ? base + (size_t)sel[0] * P.net.H : nullptr;
                dst[e] = env_ok ? base + (size_t)(s + 1) * P.net.H : nullptr;
                act[e] = sel[1];
            }
            __syncthreads();
            load_hidden_onehot(P.net, lds + P.o.X, src, act, tid);
            __syncthreads();
            mlp_recurrent_tile(P.net, P.o, lds, dst, false, nullptr, tid);
            r32 = lds[P.o.OUT + e * 4 + 0];
            v32 = lds[P.o.OUT + e * 4 + 1];
        }
        if (P.tree_mode == 2) resume = tree2_backup(smem, P, tid, env_ok, s, r32, v32, T);
        else if (a0 == 0 && env_ok) tree_expand_backup(smem, P, e, s, r32, v32);
        __syncthreads();
    }
    // the tail reads its parameters through an opaque pointer to the kernel-argument segment (see k_search_fast: merged with the loads
    // at the top of the kernel they are carried across the simulation loop as spilled scalars)
    typedef const __attribute__((address_space(4))) SearchParams* late_params_t;
    late_params_t late = (late_params_t)__builtin_amdgcn_kernarg_segment_ptr();  // (SearchParams is the kernel's only argument)
    asm volatile("" : "+s"(late));
    const SearchParams& Pt = *(const SearchParams*)late;
    if (a0 == 0 && env_ok) {
        if (Pt.tree_mode == 2) tree2_finish(smem, Pt, e, env_g);
        else tree_finish(smem, Pt, e, env_g);
    }
    if (!SCRIPTED && Pt.fuse_env && env_ok) env_step_group(Pt.fenv, env_g, a0);
}

// ---- stand-alone batched inference (network.py:62-111) on the same tile pipeline ----
struct InferParams {
    MlpNet net;
    MlpLds o;
    int t_ptr, t_pi, t_act, lds_bytes;
    int B;
    const float* in;     // obs [B][in_dim] or hidden [B][H]
    const float* const* in_ptrs;  // or per-env input rows (gather from the node store); null = dense `in`
    const int* action;   // [B] (recurrent)
    float* hidden_out;   // [B][H]
    float* const* out_ptrs;       // or per-env output rows (scatter into the node store); null = dense `hidden_out`
    float* reward;       // [B]
    float* value;        // [B]
    float* pi;           // [B][A]
};

template <bool INITIAL>
__global__ __launch_bounds__(WG_THREADS) void k_infer(const InferParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, e = tid >> 4, a0 = tid & 15;
    const int env_g = blockIdx.x * TILE_E + e;
    const bool env_ok = env_g < P.B;
    const float** src = reinterpret_cast<const float**>(smem + P.t_ptr);
    float** dst = reinterpret_cast<float**>(smem + P.t_ptr) + 16;
    float* pi = reinterpret_cast<float*>(smem + P.t_pi);
    int* act = reinterpret_cast<int*>(smem + P.t_act);
    stage_biases(P.net, lds, tid);
    if (a0 == 0) {
        src[e] = env_ok ? (P.in_ptrs ? P.in_ptrs[env_g] : P.in + (size_t)env_g * (INITIAL ? P.net.in_dim : P.net.H)) : nullptr;
        dst[e] = env_ok ? (P.out_ptrs ? P.out_ptrs[env_g] : P.hidden_out + (size_t)env_g * P.net.H) : nullptr;
        act[e] = (!INITIAL && env_ok) ? P.action[env_g] : 0;
    }
    __syncthreads();
    if (INITIAL) {
        load_obs(P.net, lds + P.o.X, src, tid);
        __syncthreads();
        mlp_initial_tile(P.net, P.o, lds, dst, pi, tid);
    } else {
        load_hidden_onehot(P.net, lds + P.o.X, src, act, tid);
        __syncthreads();
        mlp_recurrent_tile(P.net, P.o, lds, dst, true, pi, tid);
    }
    if (env_ok) {
        if (a0 == 0) {
            P.reward[env_g] = lds[P.o.OUT + e * 4 + 0];
            P.value[env_g] = lds[P.o.OUT + e * 4 + 1];
        }
        for (int a = a0; a < P.net.A; a += 16) P.pi[(size_t)env_g * P.net.A + a] = pi[e * P.net.A + a];
    }
}

}  // namespace mz
