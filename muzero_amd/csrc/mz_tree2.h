// mz_tree2.h -- tree_mode 2: the same search tree as mz_search.h's reference-order walk (bit-identical results), laid
// out and scheduled for the 16-lanes-per-env segments of the search kernels.  Included from mz_search.h.
//
//   * child ENTRY table: entry[node][action] = {vq, N, child node} of that child, kept up to date by backup.  One
//     16-byte LDS read per action lane gives everything child_Q / child_U need about a child (mcts.py:159-200); the
//     per-level dependent chain is {entry, parent N} -> {pb factor table} -> ALU, two LDS round trips.
//   * selection CACHE: every node caches its current best child {action, child, version}.  A node's pUCT ranking
//     depends only on its own N, its children's (N, vq), the root prior and the env's min-max pair.  The first three
//     change only when a backup passes through the node; a min-max change bumps the env's version and invalidates
//     all of the env's cached entries.  A level is re-evaluated at visit time on a miss, or when the cached evaluation
//     found a real tie (the tie-break draw must be consumed at visit time, like np.random.choice in mcts.py:124).
//   * BACKUP (mcts.py:129-157) runs on the env's 16 lanes: lane i owns the i-th path node from the leaf; the value
//     recurrence is a DPP shift chain in registers; W/N/Q/vq updates and the min-max reduction are lane-parallel; then
//     the best child of every path node is re-evaluated lane-parallel (16/A_pad nodes at a time) with the final
//     statistics and written to the cache.
#pragma once
// (included inside namespace mz by mz_search.h, after SearchParams / select helpers are defined)

struct __attribute__((aligned(8))) Node2 {  // 24 bytes
    double W;
    int N;
    float reward;
    short parent;
    short move;
    int player;
};
struct __attribute__((aligned(16))) Entry2 {  // 16 bytes
    double vq;  // child's reward + discount * (+/-)Q  (the min-max update value == un-normalised child_Q term)
    int cn;     // child's visit count (0: never expanded)
    int c;      // child's node index, -1 if unexpanded
};
struct SelCache {
    int packed;  // (best_action & 0xffff) | (best_child_node << 16); best_action == -1: evaluate at visit time
    int ver;
};

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v, int old = 0) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i<CTRL>((int)(b & 0xffffffffLL)), hi = dpp_i<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_SHR1 = 0x111;

// max over aligned sub-groups of `width` lanes (2, 4, 8 or 16) inside a 16-lane row
__device__ __forceinline__ float subgroup_max(float v, int width) {
    float o = dpp_f<DPP_XOR1>(v); v = o > v ? o : v;
    if (width > 2) { o = dpp_f<DPP_XOR2>(v); v = o > v ? o : v; }
    if (width > 4) { o = dpp_f<DPP_HALF_MIRROR>(v); v = o > v ? o : v; }
    if (width > 8) { o = dpp_f<DPP_MIRROR>(v); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ int row_max_i(int v) {
    int o;
    o = dpp_i<0x128>(v); v = o > v ? o : v;
    o = dpp_i<0x124>(v); v = o > v ? o : v;
    o = dpp_i<0x122>(v); v = o > v ? o : v;
    o = dpp_i<0x121>(v); v = o > v ? o : v;
    return v;
}

__device__ __forceinline__ Node2* node2_at(unsigned char* smem, const SearchParams& P, int e, int i) {
    return reinterpret_cast<Node2*>(smem + P.t2_nodes) + (e * P.NN + i);
}
__device__ __forceinline__ Entry2* entry2_row(unsigned char* smem, const SearchParams& P, int e, int i) {
    return reinterpret_cast<Entry2*>(smem + P.t2_entries) + (size_t)(e * P.NN + i) * P.A;
}
__device__ __forceinline__ SelCache* cache_at(unsigned char* smem, const SearchParams& P, int e, int i) {
    return reinterpret_cast<SelCache*>(smem + P.t_cache) + (e * P.NN + i);
}
__device__ __forceinline__ short* path_row(unsigned char* smem, const SearchParams& P, int e) {
    return reinterpret_cast<short*>(smem + P.t_path) + e * (P.NN + 3);
}
__device__ __forceinline__ int tri(int n) { return (n * (n + 1)) >> 1; }

__device__ __forceinline__ void tree2_init(unsigned char* smem, const SearchParams& P, int tid, bool env_ok, int env_g) {
    double* ft = reinterpret_cast<double*>(smem + P.t2_ftab);
    for (int i = tid; i < ((P.S + 1) * (P.S + 2)) / 2; i += WG_THREADS) ft[i] = P.ftab_tri[i];
    Entry2* en = reinterpret_cast<Entry2*>(smem + P.t2_entries);
    for (int i = tid; i < TILE_E * P.NN * P.A; i += WG_THREADS) { en[i].vq = 0.0; en[i].cn = 0; en[i].c = -1; }
    SelCache* c = reinterpret_cast<SelCache*>(smem + P.t_cache);
    for (int i = tid; i < TILE_E * P.NN; i += WG_THREADS) { c[i].packed = 0xffff; c[i].ver = -1; }  // all stale
    if (tid < TILE_E) reinterpret_cast<int*>(smem + P.t_ver)[tid] = 0;
    if ((tid & 15) == 0) {
        const int e = tid >> 4;
        Node2* r = node2_at(smem, P, e, 0);
        r->W = 0.0; r->N = 0; r->reward = 0.0f; r->parent = -1; r->move = -1; r->player = env_ok ? P.cur[env_g] : 0;
    }
}

// pUCT value of action lane `a0` of a node from its entry (child_Q + child_U, mcts.py:159-200)
__device__ __forceinline__ float puct_entry(const SearchParams& P, const Entry2& en, double f, double prior_a, double mn, double mx, bool norm,
                                            bool prior_f32) {
    float qa = 0.0f;
    if (en.cn > 0) {
        double v = en.vq;
        if (norm) v = (v - mn) / (mx - mn);
        qa = (float)v;
    }
    const float ua = prior_f32 ? ((float)prior_a * (float)f) : (float)(prior_a * f);
    return qa + ua;
}

// One descent (mcts.py:372-379).  Results segment-uniform.  All 64 lanes of every wave must call it.
__device__ __forceinline__ void tree2_select(unsigned char* smem, const SearchParams& P, int tid, bool env_ok, int env_g, int& leaf_parent,
                                             int& leaf_action) {
    const int e = tid >> 4, a0 = tid & 15, seg = (tid & 63) >> 4;
    const double* mm = reinterpret_cast<const double*>(smem + P.t_mm) + e * 2;
    int* sel = reinterpret_cast<int*>(smem + P.t_sel) + e * 4;
    short* path = path_row(smem, P, e);
    const double* ftab = reinterpret_cast<const double*>(smem + P.t2_ftab);
    const int cur_ver = reinterpret_cast<const int*>(smem + P.t_ver)[e];
    const double mn = mm[0], mx = mm[1];
    const bool norm = mx > mn, prior_f32 = (P.noise_mode == 0), lane_ok = a0 < P.A;
    const double prior_a = lane_ok ? (reinterpret_cast<const double*>(smem + P.t_prior) + e * P.A)[a0] : 0.0;
    int n = 0, cp = env_ok ? P.cur[env_g] : 0, op = env_ok ? P.opp[env_g] : 0;
    int ties = sel[3];
    bool done = !env_ok;
    int lp = 0, la = 0, lpl = 0, depth = 0;
    while (__any(!done)) {
        // round 1: everything that depends only on n
        const SelCache cc = *cache_at(smem, P, e, n);
        const int Np = node2_at(smem, P, e, n)->N;
        const Entry2 en = entry2_row(smem, P, e, n)[lane_ok ? a0 : 0];
        const int ba = (int)(short)(cc.packed & 0xffff), bc = cc.packed >> 16;
        const bool hit = (cc.ver == cur_ver) && (ba >= 0);
        if (a0 == 0 && !done) { MZ_COUNT(0, 1); MZ_COUNT(1, hit ? 1 : 0); }
        int a_sel = ba, c = bc;
        if (__any(!done && !hit)) {  // wave-uniform: some segment has to evaluate this level (best_child, mcts.py:104-127)
            const double f = ftab[tri(Np) + en.cn];  // round 2
            const float u = lane_ok ? puct_entry(P, en, f, prior_a, mn, mx, norm, prior_f32) : __uint_as_float(0xff800000u);
            const float best = butterfly16_max(u);
            const bool eq = lane_ok && (u == best);
            const unsigned long long bal = __ballot(eq);
            const unsigned msk = (unsigned)(bal >> (16 * seg)) & 0xffffu;  // tie set in ascending action order
            const int total = __popc(msk);
            int pick = 0;
            if (!done && !hit && total > 1) {  // np.random.choice consumes randomness only for a real tie
                double uu;
                if (P.rng_mode == 0) {
                    if (ties < P.max_ties) uu = P.u_tie[(size_t)env_g * P.max_ties + ties];
                    else { uu = 0.5; if (a0 == 0) atomicExch(P.err, 4); }
                } else {
                    Philox g(P.seed, P.env_offset + (unsigned)env_g, P.move_counter, 0x10000000u + (unsigned)ties);
                    uu = g.uniform();
                }
                ties++;
                pick = (int)floor(uu * (double)total);
                pick = pick >= total ? total - 1 : pick;
            }
            const int as = nth_set_bit(msk, pick);
            const int cs = row_max_i((a0 == as) ? en.c : -2);  // broadcast the chosen lane's child index
            if (!hit) { a_sel = as; c = cs; }
        }
        const int t = cp; cp = op; op = t;  // mcts.py:379
        if (!done) {
            if (a0 == 0) path[depth] = (short)n;
            depth++;
            if (c < 0 || depth > P.NN) {
                done = true;
                lp = n; la = a_sel; lpl = cp;
            } else {
                n = c;
            }
        }
    }
    if (a0 == 0 && env_ok) MZ_COUNT(2, 1);
    if (a0 == 0) {
        sel[0] = lp; sel[1] = la; sel[2] = lpl; sel[3] = ties;
        reinterpret_cast<int*>(smem + P.t_sel)[80 + e] = depth;  // expanded nodes on the path (root .. leaf parent)
    }
    leaf_parent = lp;
    leaf_action = la;
}

// expand + backup + cache refresh; executed by ALL threads (16 lanes per env cooperate); r32 / v32 segment-uniform
__device__ __forceinline__ void tree2_backup(unsigned char* smem, const SearchParams& P, int tid, bool env_ok, int s, float r32, float v32) {
    const int e = tid >> 4, a0 = tid & 15, seg = (tid & 63) >> 4;
    const int* sel = reinterpret_cast<const int*>(smem + P.t_sel) + e * 4;
    double* mm = reinterpret_cast<double*>(smem + P.t_mm) + e * 2;
    int* ver = reinterpret_cast<int*>(smem + P.t_ver) + e;
    short* path = path_row(smem, P, e);
    const int lp = sel[0], la = sel[1], cp = sel[2], nw = s + 1;
    const int depth = reinterpret_cast<const int*>(smem + P.t_sel)[80 + e];
    const int L = env_ok ? depth + 1 : 0;  // path nodes including the new one
    const double g = P.discount;
    const bool board = P.board != 0;
    if (a0 == 0 && env_ok) {  // expand (mcts.py:386); LDS operations of one wave execute in order: later reads see this
        Node2* nd = node2_at(smem, P, e, nw);
        nd->W = 0.0; nd->N = 0; nd->reward = r32; nd->parent = (short)lp; nd->move = (short)la; nd->player = cp;
        path[depth] = (short)nw;
    }
    double mn = mm[0], mx = mm[1];
    const double mn0 = mn, mx0 = mx;
    double val_in = (double)v32;
    for (int base = 0; __any(base < L); base += 16) {
        const int idx = L - 1 - (base + a0);  // lane i owns the i-th node counted from the leaf
        const bool valid = idx >= 0;
        const int p = valid ? path[idx] : 0;
        Node2* x = node2_at(smem, P, e, p);
        const double rw = valid ? (double)x->reward : 0.0;
        const bool same = valid && (x->player == cp);
        const double W0 = x->W;
        const int N0 = x->N;
        const int par = x->parent, mv = x->move;
        // value recurrence (mcts.py:152-155) as a shift chain: lane t receives lane t-1's value
        double val = val_in;
        const double prw = dpp_d<DPP_SHR1>(rw);
        const int psame = dpp_i<DPP_SHR1>(same ? 1 : 0);
        const int steps = L - base - 1;  // lanes 1..steps of this chunk hold nodes
        for (int t = 1; t < 16 && __any(t <= steps); t++) {
            const double pv = dpp_d<DPP_SHR1>(val);
            const double cand = (board && psame) ? (-prw + g * pv) : (prw + g * pv);
            if (a0 == t) val = cand;
        }
        if (valid) {
            const double W = W0 + (same ? val : -val);
            const int N = N0 + 1;
            const double Q = W / (double)N;
            const double v = board ? (rw + g * -Q) : (rw + g * Q);
            x->W = W; x->N = N;
            if (par >= 0) {
                Entry2* en = entry2_row(smem, P, e, par) + mv;
                en->vq = v; en->cn = N; en->c = p;
            }
            mx = v > mx ? v : mx;
            mn = v < mn ? v : mn;
        }
        // carry into the next chunk of 16 path nodes (deep paths only)
        const double nxt = (board && same) ? (-rw + g * val) : (rw + g * val);
        val_in = __shfl(nxt, (tid & 48) | 15, 64);
    }
    // min-max over the env's lanes (MinMaxStats.update, mcts.py:40-42)
    {
        double o;
        o = row_ror<8>(mn); mn = o < mn ? o : mn;  o = row_ror<8>(mx); mx = o > mx ? o : mx;
        o = row_ror<4>(mn); mn = o < mn ? o : mn;  o = row_ror<4>(mx); mx = o > mx ? o : mx;
        o = row_ror<2>(mn); mn = o < mn ? o : mn;  o = row_ror<2>(mx); mx = o > mx ? o : mx;
        o = row_ror<1>(mn); mn = o < mn ? o : mn;  o = row_ror<1>(mx); mx = o > mx ? o : mx;
    }
    int cur_ver = *ver;
    if (mn != mn0 || mx != mx0) { cur_ver++; if (a0 == 0 && env_ok) MZ_COUNT(3, 1); }
    if (a0 == 0 && env_ok) { mm[0] = mn; mm[1] = mx; *ver = cur_ver; }
    // refresh the best child of every node on the path with the final statistics
    const int Ap = P.A <= 2 ? 2 : (P.A <= 4 ? 4 : (P.A <= 8 ? 8 : 16));
    const int G = 16 / Ap, gi = a0 / Ap, a = a0 & (Ap - 1);
    const double* ftab = reinterpret_cast<const double*>(smem + P.t2_ftab);
    const bool norm = mx > mn, prior_f32 = (P.noise_mode == 0), lane_ok = a < P.A;
    const double prior_a = lane_ok ? (reinterpret_cast<const double*>(smem + P.t_prior) + e * P.A)[a] : 0.0;
    for (int base = 0; __any(base < L); base += G) {
        const int k = base + gi;
        const bool valid = k < L;
        const int p = valid ? path[k] : 0;
        const int Np = node2_at(smem, P, e, p)->N;
        const Entry2 en = entry2_row(smem, P, e, p)[lane_ok ? a : 0];
        const double f = ftab[tri(Np) + en.cn];
        const float u = (valid && lane_ok) ? puct_entry(P, en, f, prior_a, mn, mx, norm, prior_f32) : __uint_as_float(0xff800000u);
        const float best = subgroup_max(u, Ap);
        const bool eq = valid && lane_ok && (u == best);
        const unsigned long long bal = __ballot(eq);
        const unsigned bits = (unsigned)(bal >> (16 * seg + Ap * gi)) & ((1u << Ap) - 1u);
        const int cnt = __popc(bits);
        if (valid) {
            SelCache* cc = cache_at(smem, P, e, p);
            if (cnt == 1) {
                if (eq) { cc->packed = (a & 0xffff) | (en.c << 16); cc->ver = cur_ver; }
            } else if (a == 0) {
                cc->packed = 0xffff;  // real tie: evaluate (and draw) at visit time
                cc->ver = cur_ver;
            }
        }
    }
}

__device__ __forceinline__ void tree2_finish(unsigned char* smem, const SearchParams& P, int e, int env_g) {
    int* rv = reinterpret_cast<int*>(smem + P.t_pi0) + e * P.A;
    const Entry2* er = entry2_row(smem, P, e, 0);
    for (int a = 0; a < P.A; a++) rv[a] = er[a].cn;
    const Node2* root = node2_at(smem, P, e, 0);
    play_from_visits(smem, P, e, env_g, rv, root->W, root->N);
}

