// mz_tree2.h -- tree_mode 2: the same search tree as mz_search.h's reference-order walk (bit-identical results), laid
// out and scheduled for the 16-lanes-per-env segments of the search kernels.  Included from mz_search.h (inside
// namespace mz, after SearchParams and the select helpers).
//
//   * child ENTRY table: entry[node][action] = {vq, child_U, N, child node} of that child, kept up to date by backup.
//     One 16-byte LDS read per action lane gives everything best_child needs about a child (mcts.py:159-200): the
//     child_U term prior * f(N_parent, N_child) (mcts.py:189-200) changes only when a backup passes through the parent,
//     which recomputes it for all of the parent's actions, so selection reads no factor table and no node.
//   * NORMALISATION WITHOUT DIVIDING (MinMaxStats.normalize, mcts.py:44-48): the reference rounds
//     q64 = (v - min) / (max - min) to float32 (mcts.py:176).  Here y = (v - min) * RN(1 / (max - min)) -- the reciprocal
//     is computed once per min-max change -- differs from q64 by at most 2.5 ulp64, so float32(y) == float32(q64)
//     unless y lies within 4 ulp64 of a float32 rounding midpoint (probability ~2e-8); only then is the division
//     done.  The result is the reference's float32 bit for bit (norm_q).
//   * selection CACHE: every node caches its current best child together with the MARGIN by which it beat the
//     runner-up.  A node's pUCT ranking depends on its own N, its children's (N, vq), the root prior -- all of which
//     change only when a backup passes through the node (which refreshes the cache) -- and on the env's min-max pair,
//     which only rescales the child_Q terms.  Every min-max change k moves each normalised Q by at most
//     D_k = max(norm_new(min_old), 1 - norm_new(max_old)) (the difference of two affine maps peaks at an endpoint), so
//     a cached choice is PROVABLY still the unique argmax while 2 * (sum of D_k since it was computed) + rounding slack
//     < margin.  Otherwise (or if the cached evaluation saw a real tie, whose np.random.choice draw must be consumed at
//     visit time, mcts.py:124) the level is evaluated at visit time exactly as the reference does.  The cache is a pure
//     shortcut: it never changes a result.
//   * TWO ACTIONS (classic control): with one rival the validity of a cached choice can be tested exactly instead of through
//     the drift bound, at the cost of two FMAs per level: the best action's lead over the other one is
//         D = (Qn_best - Qn_other) + (U_best - U_other),   Qn = (vq - min) * rinv for a visited child, 0 for an unvisited one,
//     and only min and rinv change between the backups that pass through the node.  The cache keeps ud = U_best - U_other and
//     the Q part relative to a per-env reference point qref in [min, max]: both visited: a = vq_best - vq_other, k = 0; one
//     visited (sign s = +1 if it is the best one): a = s * (vq - qref), k = s; so D = (a + k * (qref - min)) * rinv + ud in
//     float32, within 3e-6 of what the evaluation would compute (every term is bounded by the range, so float32 rounding of
//     the parts stays relative to normalised units).  D > slack proves the evaluation would pick the same action without
//     a tie; 91 % of the level evaluations that the drift bound forced at 4096 CartPole envs re-confirmed the cached choice.
//   * a DESCENT is therefore mostly a chain of single 16-byte cache reads, software-pipelined: the next level's entry is
//     requested before the loop's exit test resolves (measured on gfx950 with one wave per SIMD, tools/micro/chase.hip:
//     a bare dependent ds_read_b64 chain costs 69 cycles per level, the first version of this loop -- lane predicates
//     bouncing between VALU compares and SALU mask logic, two branches -- 305, this form 180).  A segment that has
//     found its leaf parks on a SENTINEL node (index NN, margin -inf) instead of carrying a `done` predicate; the path
//     (node per depth) is written to the env's LDS path row unconditionally -- re-writing a slot with the same node
//     is harmless -- so no lane bookkeeping depends on predicates.  A normalisation switch-on (the only event that
//     invalidates every cached choice at once) resets all margins of the env to -inf: no epoch test per level.
//   * RESUMED DESCENTS: a backup refreshes the cache of every node on its path, so it knows how far the NEXT descent
//     will retrace that path: down to the first node whose refreshed choice is not "unique, valid, and the next path
//     node".  The next select starts there (the nodes above are already in the lanes' path registers) instead of at
//     the root -- in the deep, narrow trees of single-player searches most descents start at or next to the leaf.
//   * BACKUP (mcts.py:129-157) runs on the env's 16 lanes: lane i owns the i-th path node from the leaf; the value
//     recurrence is a DPP shift chain in registers; W/N/Q/vq updates, the min-max reduction and the best-child refresh
//     of every path node (each lane loops over its node's actions) are lane-parallel.
#pragma once


struct __attribute__((aligned(16))) Node2 {  // 16 bytes: one ds_read_b128
    double W;
    float reward;
    short N;
    unsigned short link;  // bits 0-7 parent node + 1 (0: the root), bits 8-11 the move that led here, bits 12-15 player to move
};
__device__ __forceinline__ unsigned short node_link(int parent, int move, int player) {
    return (unsigned short)((parent + 1) | (move << 8) | (player << 12));
}
struct __attribute__((aligned(16))) Entry2 {  // 16 bytes
    double vq;  // child's reward + discount * (+/-)Q  (the min-max update value == un-normalised child_Q term)
    float U;    // child_U of this action for the parent's and child's current visit counts (float32, mcts.py:189-200)
    short cn;   // child's visit count (0: never expanded)
    short c;    // child's node index, -1 if unexpanded
};
struct __attribute__((aligned(16))) SelCache {  // 16 bytes: one ds_read_b128
    int packed;  // bits 16-31: where a descent that follows this entry goes next -- the best child's node, or the SENTINEL slot (index NN)
                 // if that child is not expanded yet (the descent stops there: one select per level instead of two in the chase);
                 // bits 0-7 best action (0xff: evaluate at visit time), bits 8-15 env epoch when computed (mod 256,
                 // bumped when normalisation switches on), bits 16-31 best child node (signed, -1 unexpanded)
    float t;     // A > 2: margin + 2 * drift at compute time, rounded down: valid while 2 * drift_now + slack < t
                 // A == 2: ud = child_U(best) - child_U(other)
    float a, k;  // A == 2 (see "TWO ACTIONS" in the header): the best action's lead is (a + k * (qref - min)) * rinv + ud
};
// Per-env search state that select and backup hand to each other IN REGISTERS (identical in the env's 16 lanes: every
// update is computed segment-uniformly) -- through LDS each hand-over cost a write, a read and their round trips, ~1 k
// cycles per simulation for a wave that runs alone on its SIMD.
struct Tree2Env {
    double mn, mx;   // MinMaxStats (mcts.py:36-48)
    double drift;    // A > 2: sum of D_k
    double rinv;     // A > 2: RN(1 / (max - min)) of the current min-max pair (valid while max > min)
    double qref;     // A == 2: the env's minimum when normalisation switched on (any fixed point of [min, max] would do)
    int epoch, ties; // normalisation switch-ons; tie draws consumed so far (indexes the tie RNG stream, mcts.py:124)
    int lp, la, cp, depth;  // the last descent: leaf parent, leaf action, player to move at the leaf, expanded nodes on the path
};
__device__ __forceinline__ void tree2_env_init(Tree2Env& t, const SearchParams& P) {
    t.mn = P.has_bounds ? P.kb_min : __longlong_as_double(0x7ff0000000000000LL);  // MinMaxStats, mcts.py:36-38
    t.mx = P.has_bounds ? P.kb_max : __longlong_as_double(0xfff0000000000000LL);
    t.drift = 0.0;
    t.rinv = P.has_bounds ? 1.0 / (P.kb_max - P.kb_min) : 0.0;
    t.qref = P.has_bounds ? P.kb_min : 0.0;
    t.epoch = 0; t.ties = 0; t.lp = 0; t.la = 0; t.cp = 0; t.depth = 0;
}

// float32(MinMaxStats.normalize(v)) exactly as the reference computes it, usually without the division (see header).
// Straight-line: the division sits behind a wave-level test that is almost never true.
__device__ __forceinline__ float norm_q(double v, double mn, double mx, double rinv) {
    const double d = v - mn;
    double y = d * rinv;
    const long long b = __double_as_longlong(y);
    const int lo = (int)(b & 0x1fffffffLL) - 0x10000000;
    const int ex = (int)((b >> 52) & 0x7ff);
    const bool amb = ((lo >= -4) & (lo <= 4)) | ((ex < 1023 - 120) & (d != 0.0));  // ambiguous rounding / float32 subnormal range
    if (__builtin_amdgcn_ballot_w64(amb) != 0) {
        const double exact = d / (mx - mn);
        y = amb ? exact : y;
    }
    return (float)y;
}

// child_U of one action (mcts.py:189-200): float64 product rounded once in self-play; float32 * float32 when the root
// prior stayed float32 (deterministic search under numpy 2)
__device__ __forceinline__ float child_u(double prior_a, double f, bool prior_f32) {
    // both forms are computed and SELECTED: written as a conditional expression hipcc turns the (wave-uniform) choice into a branch
    // around each form -- one wave-level branch per action in the backup's refresh loops, ~40 cycles each for a wave that runs alone
    float u32 = (float)prior_a * (float)f, u64 = (float)(prior_a * f);
    asm("" : "+v"(u32), "+v"(u64));
    return prior_f32 ? u32 : u64;
}
// two-action lead test (header, "TWO ACTIONS"): the float32 reciprocal of the range; select and backup must use this very
// expression -- the backup's resume point promises what the next descent's test will say
__device__ __forceinline__ float lead_r32(double mn, double mx) {
    // NaN (the test then fails and the level is evaluated) while normalisation is off, and for ranges so small that the
    // float32 parts of the test would leave the normal range
    const float rg = (float)(mx - mn);
    return rg > 1e-30f ? __builtin_amdgcn_rcpf(rg) : __uint_as_float(0x7fc00000u);
}
constexpr float kCacheSlack = 4e-5f;  // float32 rounding of child_Q / child_U / their sum, for |ucb| < 64

// (bound_ctrl: lanes without a source lane -- lane 0 of a row under row_shr -- read 0, which is what `old = 0` gave them, without the
// v_mov that pre-loads the destination)
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i<CTRL>((int)(b & 0xffffffffLL)), hi = dpp_i<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int DPP_SHR1 = 0x111;

__device__ __forceinline__ int row_max_i(int v) {  // signed max over the 16 lanes of a row (see butterfly16_max)
    asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(v));
    return v;
}

__device__ __forceinline__ Node2* node2_at(unsigned char* smem, const SearchParams& P, int e, int i) {
    return reinterpret_cast<Node2*>(smem + P.t2_nodes) + (e * P.NN + i);
}
__device__ __forceinline__ Entry2* entry2_row(unsigned char* smem, const SearchParams& P, int e, int i) {
    return reinterpret_cast<Entry2*>(smem + P.t2_entries) + (size_t)(e * P.NN + i) * P.A;
}
__device__ __forceinline__ SelCache* cache_at(unsigned char* smem, const SearchParams& P, int e, int i) {
    return reinterpret_cast<SelCache*>(smem + P.t_cache) + (e * (P.NN + 1) + i);  // slot NN of every env: the sentinel
}
__device__ __forceinline__ short* path_row(unsigned char* smem, const SearchParams& P, int e) {
    return reinterpret_cast<short*>(smem + P.t_path) + e * (P.NN + 3);
}
__device__ __forceinline__ int tri(int n) { return (n * (n + 1)) >> 1; }

__device__ __forceinline__ void tree2_init(unsigned char* smem, const SearchParams& P, int tid, bool env_ok, int env_g) {
    double* ft = reinterpret_cast<double*>(smem + P.t2_ftab);
    for (int i = tid; i < ((P.S + 1) * (P.S + 2)) / 2; i += WG_THREADS) ft[i] = P.ftab_tri[i];
    Entry2* en = reinterpret_cast<Entry2*>(smem + P.t2_entries);
    // U = prior * f(0, 0) = 0 for the root before its first visit (sqrt(0), mcts.py:193-195); every other node's row is
    // written by the backup that creates it
    for (int i = tid; i < TILE_E * P.NN * P.A; i += WG_THREADS) { en[i].vq = 0.0; en[i].U = 0.0f; en[i].cn = 0; en[i].c = -1; }
    SelCache* c = reinterpret_cast<SelCache*>(smem + P.t_cache);
    for (int i = tid; i < TILE_E * (P.NN + 1); i += WG_THREADS) { c[i].packed = 0xffff; c[i].t = __uint_as_float(0xff800000u); c[i].a = __uint_as_float(0xff800000u); c[i].k = 0.0f; }  // -inf: never a hit
    if ((tid & 15) == 0) {
        const int e = tid >> 4;
        Node2* r = node2_at(smem, P, e, 0);
        r->W = 0.0; r->N = 0; r->reward = 0.0f; r->link = node_link(-1, 0, env_ok ? P.cur[env_g] : 0);
    }
}

// pUCT value of one action of a node from its entry (child_Q + child_U, mcts.py:159-200), exact
__device__ __forceinline__ float puct_entry(const Entry2& en, double mn, double mx, double rinv, bool norm) {
    // straight-line: lanes whose child is unvisited compute a value too and drop it (a branch on `cn > 0` costs more)
    const float qn = norm_q(en.vq, mn, mx, rinv), qr = (float)en.vq;
    const float qa = en.cn > 0 ? (norm ? qn : qr) : 0.0f;
    return qa + en.U;
}

// One descent (mcts.py:372-379).  Results segment-uniform.  All 64 lanes of every wave must call it.
// The nodes visited at depths 0 .. k-1 end up in the env's LDS path row (path[d] = node at depth d).
// Two alternating phases: (A) the pointer chase along decided cache entries; (B) when no segment of the wave can
// advance by its cache, one full evaluation of the current level for the segments that have not found their leaf
// (best_child, mcts.py:104-127).
// `resume` (from the previous tree2_backup; 0 for the first descent): node to start at | its depth << 16 -- the path row
// still holds the nodes above it.  `cp0` / `op0`: the env's current / opponent player at the root.
// AM: what the caller knows about the action count at compile time (-1 nothing, 2 exactly two actions, 0 more than two) --
// the benchmark kernels fix it, which removes the other path's code and scalars from their simulation loop
template <bool PUBLISH = false, int AM = -1>
__device__ __forceinline__ void tree2_select(unsigned char* smem, const SearchParams& P, int tid, bool env_ok, int env_g, Tree2Env& T, int cp0,
                                             int op0, int resume = 0) {
    const int e = tid >> 4, a0 = tid & 15, seg = (tid & 63) >> 4;
    short* path = path_row(smem, P, e);
    const SelCache* cb = cache_at(smem, P, e, 0);
    const Tree2Env& st = T;
    const float thr = 2.0f * __double2float_ru(st.drift) + kCacheSlack;
    const double mn = T.mn, mx = T.mx;
    const bool norm = mx > mn, lane_ok = a0 < P.A, two = AM < 0 ? P.A == 2 : AM == 2;
    float r32 = lead_r32(mn, mx);  // two actions: exact lead test (header)
    // (kept a VALUE: left alone hipcc splits lead_r32's `range > 1e-30 ? rcp : NaN` into a mask that it ANDs with every level's
    // compare -- a VALU -> SALU -> VALU hop per level of the chase; as a NaN the compare alone says no)
    asm volatile("" : "+v"(r32));
    const float dmp = (float)(st.qref - mn);
    const int SENT = P.NN;  // cache slot NN of every env: margin -inf, never advances
    int n = env_ok ? (resume & 0xffff) : SENT, ties = T.ties;
    int lp = 0, la = 0, k = resume >> 16;  // k: levels descended so far
    MZ_TS_DECL
    MZ_TS_START();
    for (;;) {
        // ---- phase A: chase cached best children (software-pipelined: see the header) ----
        MZ_TS(0);  // [0] prelude / between phases
        auto chase = [&](auto two_tag) {  // two copies of the loop: the exact two-action test costs the margin scheme's levels two FMAs
            constexpr bool TWO = decltype(two_tag)::value;
            SelCache cc = cb[n];
            for (;;) {
                MZ_TS_COUNT(1);  // [1] phase-A iterations
                // (while normalisation is off r32 is NaN, and entries stored then carry a = -inf: the test fails)
                const bool adv = TWO ? (fmaf(fmaf(cc.k, dmp, cc.a), r32, cc.t) > kCacheSlack) : (thr < cc.t);
                const int c = cc.packed >> 16;  // the best child, or SENT if it is unexpanded (the writers substitute it)
                const bool stop = adv & (c == SENT);
                const int nxt = adv ? c : n;
                const SelCache nc = cb[nxt];
                if (a0 == 0) path[k] = (short)n;  // node at depth k (parked segments re-write their spare slot)
                lp = stop ? n : lp;
                la = stop ? (cc.packed & 0xff) : la;
                k += adv ? 1 : 0;
                if (__builtin_amdgcn_ballot_w64(adv) == 0) break;
                n = nxt;
                cc = nc;
            }
        };
        if (two) chase(std::true_type{});
        else chase(std::false_type{});
        MZ_TS(2);  // [2] phase-A cycles
        if (__builtin_amdgcn_ballot_w64(n != SENT) == 0) break;
        MZ_TS_COUNT(3);  // [3] phase-B rounds
        // ---- phase B: every segment that is not parked sits on a level its cache cannot decide: evaluate it ----
        {
            const bool live = n != SENT;
            int as, cs;
            if (two) {
                // TWO actions: every lane evaluates action (lane & 1) and takes the other one's value and child from its
                // neighbour (one quad permute each) -- no segment reduction, no ballot, no child broadcast
                const Entry2 en = entry2_row(smem, P, e, live ? n : 0)[a0 & 1];
                // two-action searches keep no reciprocal: evaluations are rare.  (The empty asm pins the division inside this
                // rarely taken block: left alone, hipcc hoists the loop-invariant quotient to the top of every descent.)
                double range = mx - mn;
                asm volatile("" : "+v"(range));
                const double rinv2 = norm ? 1.0 / range : 0.0;
                const float u = puct_entry(en, mn, mx, rinv2, norm);
                MZ_TS(4);  // [4] phase B: entry read + pUCT value
                const float uo = __int_as_float(dpp_i<0xb1>(__float_as_int(u)));  // quad_perm [1,0,3,2]
                const int co = dpp_i<0xb1>((int)en.c);
                const bool odd = (a0 & 1) != 0;
                const float u0 = odd ? uo : u, u1 = odd ? u : uo;
                const int c0 = odd ? co : (int)en.c, c1 = odd ? (int)en.c : co;
                as = u1 > u0 ? 1 : 0;  // np.where(ucb == max)[0]: the first maximum unless they tie
                const bool tie = live & (u0 == u1);
                if (__any(tie)) {
                    if (tie) {  // np.random.choice consumes randomness only for a real tie
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
                        const int pick = (int)floor(uu * 2.0);
                        as = pick >= 2 ? 1 : pick;
                    }
                }
                MZ_TS(5);  // [5] phase B: max, tie set, (rare) draw
#ifdef MZ_COUNTERS
                const int ocn = dpp_i<0xb1>((int)en.cn);
                if (live && a0 == 0) {  // why was this level evaluated?  [4] evaluations, [5] both children visited, [6] of those: the cached choice still stands, [7] one child unvisited and the cached choice still stands
                    const SelCache pc = cb[n];
                    const int ca = pc.packed & 0xff;
                    const bool both = (en.cn > 0) && (ocn > 0);
                    MZ_COUNT(4, 1);
                    if (both) MZ_COUNT(5, 1);
                    if (both && ca != 0xff && ca == as && !tie) MZ_COUNT(6, 1);
                    if (!both && ca != 0xff && ca == as && !tie) MZ_COUNT(7, 1);
                }
#endif
                cs = as ? c1 : c0;
                MZ_TS(6);  // [6] phase B: pick + child broadcast
            } else {
            const Entry2 en = entry2_row(smem, P, e, live ? n : 0)[lane_ok ? a0 : 0];
            const float u = lane_ok ? puct_entry(en, mn, mx, st.rinv, norm) : __uint_as_float(0xff800000u);
            MZ_TS(4);  // [4] phase B: entry read + pUCT value
            const float best = butterfly16_max(u);
            const bool eq = lane_ok & (u == best);
            const unsigned long long bal = __ballot(eq);
            const unsigned msk = (unsigned)(bal >> (16 * seg)) & 0xffffu;  // tie set in ascending action order
            const int total = __popc(msk);
            int pick = 0;
            if (__any(live & (total > 1))) {
                if (live && total > 1) {  // np.random.choice consumes randomness only for a real tie
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
            }
            MZ_TS(5);  // [5] phase B: max, tie set, (rare) draw
            as = nth_set_bit(msk, pick);
            cs = row_max_i((a0 == as) ? (int)en.c : -2);  // broadcast the chosen lane's child index
            MZ_TS(6);  // [6] phase B: pick + child broadcast
            }
            if (live && a0 == 0) path[k] = (short)n;
            const bool stop = live & (cs < 0);
            lp = stop ? n : lp;
            la = stop ? as : la;
            k += live ? 1 : 0;
            n = live ? (cs < 0 ? SENT : cs) : n;
        }
        MZ_TS(7);  // [7] phase B: bookkeeping
    }
    if (a0 == 0 && env_ok) MZ_COUNT(2, 1);
    // players swap at every level (mcts.py:379): the leaf's player follows from the parity of the depth
    T.lp = lp; T.la = la; T.cp = (k & 1) ? op0 : cp0; T.ties = ties; T.depth = k;  // k: expanded nodes on the path (root .. leaf parent)
    if (PUBLISH && a0 == 0) {  // kernels whose network evaluation reads the leaf from LDS
        int* sel = reinterpret_cast<int*>(smem + P.t_sel) + e * 4;
        sel[0] = lp; sel[1] = la;
    }
    MZ_TS(8);  // [8] epilogue (sel[] writes)
    MZ_TS_COUNT(9);  // [9] calls
    MZ_TS_FLUSH(0);
}

// What the backup needs of the tree BEFORE the network's outputs exist: this lane's path node (first chunk of 16 path positions)
// and that node's record.  The path is known since the select, nothing writes the tree during the network evaluation, so the two
// dependent LDS reads (path row -> node) can be issued ahead of the last barrier of the evaluation and of the value row's softmax
// instead of in front of the value recurrence (round 2's stamps: "expand + loads" ~1 k cycles per backup).
struct Backup2Pre {
    int p;      // node at path position L - 1 - a0 (the new node for a0 == 0; 0 for lanes beyond the path)
    Node2 xn;   // its record (the new node's is built by the backup itself)
    // two-action searches: the node's refreshed child_U pair and what pass 2 needs of its two entries.  child_U depends on visit
    // counts only -- known before the simulation's value is: the node's own N + 1, its path child's N + 1 (from the neighbour lane),
    // the sibling's as stored -- so the factor-table / prior reads, the two products and the U stores are done here too; of the
    // entries only the path child's value changes later, and that arrives in pass 2 from the neighbour lane's registers (DPP),
    // not through an LDS write -> read round trip.
    float ua0, ua1;
    double vq0, vq1;  // entries' Q terms as stored (the path child's is replaced in pass 2)
    int cn0, cn1;     // visit counts AFTER this backup
    int c0, c1;       // child node indices as stored (the path child's slot of a fresh expansion is filled in pass 2)
    int mvc;          // action of this node's path child (lane a0 - 1), -1: none (the new node; lanes beyond the path)
};
template <int AM>
__device__ __forceinline__ Backup2Pre tree2_backup_prefetch(unsigned char* smem, const SearchParams& P, int tid, bool env_ok, int s, const Tree2Env& T) {
    const int e = tid >> 4, a0 = tid & 15;
    const short* path = path_row(smem, P, e);
    const int L = env_ok ? T.depth + 1 : 0, idx = L - 1 - a0;
    const int from_row = (int)path[idx >= 0 ? idx : 0];
    Backup2Pre pre;
    pre.p = idx < 0 ? 0 : (idx == L - 1 ? s + 1 : from_row);
    pre.xn = *node2_at(smem, P, e, idx == L - 1 ? 0 : pre.p);  // (the new node's slot holds nothing yet: any valid address)
    if constexpr (AM == 2) {
        const bool is_new = idx == L - 1;
        const int mv_self = is_new ? T.la : ((pre.xn.link >> 8) & 15), n_after = (is_new ? 0 : (int)pre.xn.N) + 1;
        const int mvn = dpp_i<DPP_SHR1>(mv_self), cnn = dpp_i<DPP_SHR1>(n_after);  // the neighbour lane's node is this node's path child
        const bool has_child = (a0 >= 1) & (idx >= 0);
        pre.mvc = has_child ? mvn : -1;
        Entry2* er = entry2_row(smem, P, e, pre.p);
        const Entry2 e0 = er[0], e1 = er[1];
        pre.cn0 = pre.mvc == 0 ? cnn : (int)e0.cn;
        pre.cn1 = pre.mvc == 1 ? cnn : (int)e1.cn;
        const double* frow = reinterpret_cast<const double*>(smem + P.t2_ftab) + tri(n_after);
        const double* prior = reinterpret_cast<const double*>(smem + P.t_prior) + e * P.A;
        const double f0 = frow[pre.cn0], f1 = frow[pre.cn1], p0 = prior[0], p1 = prior[1];
        const bool prior_f32 = (P.noise_mode == 0 && !P.legacy_promo);
        if (prior_f32) {  // (one wave-uniform branch for the pair, not two forms per product)
            pre.ua0 = (float)p0 * (float)f0;
            pre.ua1 = (float)p1 * (float)f1;
        } else {
            pre.ua0 = (float)(p0 * f0);
            pre.ua1 = (float)(p1 * f1);
        }
        if (idx >= 0) { er[0].U = pre.ua0; er[1].U = pre.ua1; }
        pre.vq0 = e0.vq; pre.vq1 = e1.vq;
        pre.c0 = e0.c; pre.c1 = e1.c;
    }
    return pre;
}

// expand + backup + cache refresh; executed by ALL threads (16 lanes per env cooperate); r32 / v32 segment-uniform;
// the path row as written by tree2_select of the same simulation
// returns the resume point of the next descent (see header)
// ACT: the action count when the caller knows it at compile time (0: not known): pass 2 fully unrolled, all LDS reads of a node's
// refresh in two batches (the generic loop's rounds of four paid two exposed LDS round trips and two wave-level branches each:
// 4.7 k cycles per backup at ten actions)
template <int AM = -1, int ACT = 0, bool PRE = false>
__device__ __forceinline__ int tree2_backup(unsigned char* smem, const SearchParams& P, int tid, bool env_ok, int s, float r32, float v32,
                                            Tree2Env& T, const Backup2Pre* pre = nullptr) {
    const int e = tid >> 4, a0 = tid & 15;
    const bool two = AM < 0 ? P.A == 2 : AM == 2;
    const short* path = path_row(smem, P, e);
    const double* ftab = reinterpret_cast<const double*>(smem + P.t2_ftab);
    const double* prior = reinterpret_cast<const double*>(smem + P.t_prior) + e * P.A;
    const int lp = T.lp, la = T.la, cp = T.cp, nw = s + 1;
    const int L = env_ok ? T.depth + 1 : 0;  // path nodes including the new one
    const double g = P.discount;
    const bool board = AM == 2 ? false : P.board != 0;  // (the launcher picks AM == 2 for single-player searches only)
    Tree2Env& st = T;
    double v_first = 0.0;     // this lane's node's refreshed Q term (first chunk), for the parent lane's pass 2 (PRE)
    int n_after = 0;          // this lane's node's visit count after pass 1 (single-chunk paths: pass 2 need not re-read it)
    int p_first = 0;          // this lane's node in the first chunk of 16 path positions (pass 2 need not re-read the path row)
    MZ_TS_DECL
    MZ_TS_START();
#ifdef MZ_EXP_PAD  // diagnostic builds: MZ_EXP_PAD dependent VALU instructions of dead work in front of the backup (is this phase on the critical path?)
    {
        float pad = (float)tid;
#pragma unroll
        for (int i = 0; i < MZ_EXP_PAD; i++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(pad));
        if (pad == 12345.678f) atomicExch(P.err, 99);
    }
#endif
    if (a0 == 0 && env_ok) {  // expand (mcts.py:386); LDS operations of one wave execute in order: later reads see this
        Node2* nd = node2_at(smem, P, e, nw);
        nd->W = 0.0; nd->N = 0; nd->reward = r32; nd->link = node_link(lp, la, cp);
    }
    double mn = T.mn, mx = T.mx;
    const double mn0 = mn, mx0 = mx;
    double val_in = (double)v32;
    // pass 1: statistics (lane i owns the i-th node counted from the leaf)
    for (int base = 0; __any(base < L); base += 16) {
        const int idx = L - 1 - (base + a0);
        const bool valid = idx >= 0;
        // node at path position idx: the new node, or the LDS path row (unconditional loads, selects afterwards: a load
        // under a lane predicate becomes a branch with its own LDS round trip)
        int p;
        Node2 xn;
        if (PRE && base == 0) {  // (compile-time + wave-uniform) prefetched before the network's outputs existed
            p = pre->p;
            xn = pre->xn;
            if (idx == L - 1) { xn.W = 0.0; xn.N = 0; xn.reward = r32; xn.link = node_link(lp, la, cp); }  // the node expanded above
        } else {
            const int from_row = (int)path[idx >= 0 ? idx : 0];
            p = !valid ? 0 : (idx == L - 1 ? nw : from_row);
            xn = *node2_at(smem, P, e, p);
        }
        if (base == 0) p_first = p;
        Node2* x = node2_at(smem, P, e, p);
        const float rwf = xn.reward;
        const double W0 = xn.W;
        const int N0 = xn.N;
        const int lk = xn.link, par = (lk & 0xff) - 1, mv = (lk >> 8) & 15, pl = lk >> 12;
        const double rw = valid ? (double)rwf : 0.0;
        const bool same = valid & (pl == cp);
        MZ_TS(0);  // [0] backup: expand + path node loads
        // value recurrence (mcts.py:152-155) as a shift chain: lane t receives lane t-1's value
        double val = val_in;
        const double prw = dpp_d<DPP_SHR1>(rw);
        const int psame = dpp_i<DPP_SHR1>(same ? 1 : 0);
        const int steps = L - base - 1;  // lanes 1..steps of this chunk hold nodes
        // (the term a lane adds is fixed: fold the sign choice once; exit test every 4 levels only -- a wave-level branch per
        // level costs as much as the level's arithmetic)
        const double sprw = (board && psame) ? -prw : prw;
        for (int t0 = 1; t0 < 16 && __any(t0 <= steps); t0 += 4) {
#pragma unroll
            for (int t = t0; t < t0 + 4; t++) {
                const double pv = dpp_d<DPP_SHR1>(val);
                const double cand = sprw + g * pv;
                val = (a0 == t) ? cand : val;
            }
        }
        MZ_TS(1);  // [1] backup: value chain
        if (valid) {
            const double W = W0 + (same ? val : -val);
            const int N = N0 + 1;
            const double Q = W / (double)N;
            const double v = board ? (rw + g * -Q) : (rw + g * Q);
            x->W = W; x->N = (short)N;
            n_after = N;
            if (base == 0) v_first = v;
            if (par >= 0) {
                Entry2* en = entry2_row(smem, P, e, par) + mv;
                en->vq = v; en->cn = (short)N; en->c = (short)p;
            }
            mx = v > mx ? v : mx;
            mn = v < mn ? v : mn;
        }
        MZ_TS(2);  // [2] backup: W / N / Q update, entry write
        // carry into the next chunk of 16 path nodes (deep paths only)
        const double nxt = (board && same) ? (-rw + g * val) : (rw + g * val);
        val_in = __shfl(nxt, (tid & 48) | 15, 64);
    }
    // min-max over the env's lanes (MinMaxStats.update, mcts.py:40-42).  Most backups move neither bound (known bounds: almost none
    // does; without them the range settles after the first simulations): one wave-level test skips the two float64 butterflies --
    // every lane then still holds the old pair
    if (__builtin_amdgcn_ballot_w64((mx > mx0) | (mn < mn0)) != 0) {
        double o;
        o = row_ror<8>(mn); mn = o < mn ? o : mn;  o = row_ror<8>(mx); mx = o > mx ? o : mx;
        o = row_ror<4>(mn); mn = o < mn ? o : mn;  o = row_ror<4>(mx); mx = o > mx ? o : mx;
        o = row_ror<2>(mn); mn = o < mn ? o : mn;  o = row_ror<2>(mx); mx = o > mx ? o : mx;
        o = row_ror<1>(mn); mn = o < mn ? o : mn;  o = row_ror<1>(mx); mx = o > mx ? o : mx;
    }
    MZ_TS(3);  // [3] backup: min-max reduction
    // cache-validity bookkeeping for the min-max change of this backup (see header)
    bool switched_on = false;
    if (mn != mn0 || mx != mx0) {
        switched_on = !(mx0 > mn0);
        if (two) {
            // exact lead test: no drift, and the reciprocal is formed where it is needed
        } else if (mx0 > mn0) {
            st.rinv = 1.0 / (mx - mn);  // the one division per min-max change (norm_q)
            // D_k is only an upper bound: the products are within 2 ulp of the quotients, the 1.000001 factor covers that
            const double d_lo = (mn0 - mn) * st.rinv, d_hi = (mx - mx0) * st.rinv;
            st.drift += (d_lo > d_hi ? d_lo : d_hi) * 1.000001 + 1e-12;
        } else {
            st.rinv = mx > mn ? 1.0 / (mx - mn) : 0.0;
        }
        if (switched_on) {
            st.epoch++;  // normalisation may switch on: nothing cached before survives
            st.qref = mn;
        }
        if (a0 == 0 && env_ok) MZ_COUNT(3, 1);
    }
    T.mn = mn; T.mx = mx;
    if (__any(switched_on)) {
        // normalisation may have switched on: no cached choice of this env survives (rare: once per search); the path
        // nodes get fresh entries in pass 2 below
        if (switched_on) {
            SelCache* cbase = cache_at(smem, P, e, 0);
            for (int i = a0; i < P.NN; i += 16) { cbase[i].t = __uint_as_float(0xff800000u); cbase[i].a = __uint_as_float(0xff800000u); }
        }
    }
    MZ_TS(4);  // [4] backup: drift / rinv bookkeeping + stores
    // pass 2: child_U and best child of every path node with the final statistics (same lane ownership; LDS ops are in
    // order, so the entry writes of pass 1 -- all from this wave -- are visible)
    const bool norm = mx > mn, prior_f32 = (P.noise_mode == 0 && !P.legacy_promo);
    const float thr = 2.0f * __double2float_ru(st.drift) + kCacheSlack;  // what the next select will compare the margins with
    int resume = 0;
    for (int base = 0; __any(base < L); base += 16) {
        const int idx = L - 1 - (base + a0);
        const bool valid = idx >= 0;
        int p = p_first;  // (the path row is read again only for paths deeper than 16: the load sat in front of the whole pass)
        if (base > 0) {   // wave-uniform
            const int from_row = (int)path[idx >= 0 ? idx : 0];
            p = !valid ? 0 : (idx == L - 1 ? nw : from_row);
        }
        bool decided = false;  // the refreshed cache entry will let the next descent pass through p without evaluating it
        int bestc = -1;
        if (PRE && AM == 2 && base == 0) {
            // two actions, first chunk: everything but the path child's new value was prepared by tree2_backup_prefetch; that value
            // and the child's node index come from the neighbour lane's registers.  No LDS read in this pass.
            const double vc = dpp_d<DPP_SHR1>(v_first);
            const int pc = dpp_i<DPP_SHR1>(p);
            const int mvc = pre->mvc;
            const double vq0 = mvc == 0 ? vc : pre->vq0, vq1 = mvc == 1 ? vc : pre->vq1;
            const int c0 = mvc == 0 ? pc : pre->c0, c1 = mvc == 1 ? pc : pre->c1;
            const float ua0 = pre->ua0, ua1 = pre->ua1;
            const bool v0 = pre->cn0 > 0, v1 = pre->cn1 > 0;
            const float r32n = lead_r32(mn, mx), dmpn = (float)(st.qref - mn);
            const double x0 = v0 ? vq0 : st.qref, x1 = v1 ? vq1 : st.qref;
            const float a10 = (float)(x1 - x0), k10 = (float)((v1 ? 1 : 0) - (v0 ? 1 : 0)), ud10 = ua1 - ua0;
            const float d10 = fmaf(fmaf(k10, dmpn, a10), r32n, ud10);
            const bool b1 = d10 > 0.0f;
            bestc = b1 ? c1 : c0;
            MZ_TS(5);  // [5] backup pass 2: per-action loop
            if (valid) {
                SelCache cc;
                cc.packed = (b1 ? 1 : 0) | ((st.epoch & 0xff) << 8) | ((bestc < 0 ? P.NN : bestc) << 16);
                cc.t = b1 ? ud10 : -ud10;
                cc.a = norm ? (b1 ? a10 : -a10) : __uint_as_float(0xff800000u);
                cc.k = b1 ? k10 : -k10;
                *cache_at(smem, P, e, p) = cc;
                decided = norm & (fabsf(d10) > kCacheSlack);
            }
        } else
        if (valid) {
            const int Np = L <= 16 ? n_after : (int)node2_at(smem, P, e, p)->N;
            const double* frow = ftab + tri(Np);
            Entry2* er = entry2_row(smem, P, e, p);
            float best = __uint_as_float(0xff800000u), second = __uint_as_float(0xff800000u);
            int besta = 0, cnt = 0;
            // the cached ranking's Q term (vq - min) * rinv, or vq itself while normalisation is off, as ONE expression: (vq - 0) * 1 is vq
            // exactly, and a per-action select between the two forms compiles to a lane-level branch per action
            const double q_sub = norm ? mn : 0.0, q_mul = norm ? st.rinv : 1.0;
            // CH actions per round (their LDS reads overlap)
            auto rank_actions = [&](auto ch_tag) {
                constexpr int CH = decltype(ch_tag)::value;
                for (int a4 = 0; a4 < P.A; a4 += CH) {
                    Entry2 en[CH];
                    double f[CH], pr[CH];
#pragma unroll
                    for (int j = 0; j < CH; j++) en[j] = er[a4 + j < P.A ? a4 + j : P.A - 1];
#pragma unroll
                    for (int j = 0; j < CH; j++) { f[j] = frow[en[j].cn]; pr[j] = prior[a4 + j < P.A ? a4 + j : P.A - 1]; }
#pragma unroll
                    for (int j = 0; j < CH; j++) {
                        const int a = a4 + j;
                        if (a < P.A) {  // wave-uniform
                            const float ua = child_u(pr[j], f[j], prior_f32);  // this node's N (and one child's) just changed
                            er[a].U = ua;
                            // the cached ranking may use the un-checked product: its error (1 ulp of float32) is far inside
                            // the cache's slack, and a choice that is not decided by more than the slack is re-evaluated
                            // exactly at visit time anyway
                            const float qn = (float)((en[j].vq - q_sub) * q_mul);
                            const float u = (en[j].cn > 0 ? qn : 0.0f) + ua;
                            // first maximum, number of actions tied with it, runner-up -- as selects (branches on lane
                            // predicates cost a wave that runs alone on its SIMD far more than the selects do)
                            const bool gt = u > best, eq = u == best;
                            second = gt ? best : ((!eq & (u > second)) ? u : second);
                            cnt = gt ? 1 : (eq ? cnt + 1 : cnt);
                            besta = gt ? a : besta;
                            bestc = gt ? (int)en[j].c : bestc;
                            best = gt ? u : best;
                        }
                    }
                }
            };
            if (two) {
                // two actions: candidate = first maximum of the (un-checked) ranking; whether the next descents may follow it
                // is decided by the exact lead test (header, "TWO ACTIONS"), here and at every visit
                const Entry2 e0 = er[0], e1 = er[1];
                const double f0 = frow[e0.cn], f1 = frow[e1.cn], p0 = prior[0], p1 = prior[1];
                const float ua0 = child_u(p0, f0, prior_f32), ua1 = child_u(p1, f1, prior_f32);  // this node's N (and one child's) just changed
                er[0].U = ua0; er[1].U = ua1;
                // the lead of action 1 over action 0 (header, "TWO ACTIONS"); its sign IS the ranking, so no separate one is made
                const bool v0 = e0.cn > 0, v1 = e1.cn > 0;
                const float r32n = lead_r32(mn, mx), dmpn = (float)(st.qref - mn);
                const double x0 = v0 ? e0.vq : st.qref, x1 = v1 ? e1.vq : st.qref;  // an unvisited child's Q term is 0: it enters through k
                const float a10 = (float)(x1 - x0), k10 = (float)((v1 ? 1 : 0) - (v0 ? 1 : 0)), ud10 = ua1 - ua0;
                const float d10 = fmaf(fmaf(k10, dmpn, a10), r32n, ud10);
                const bool b1 = d10 > 0.0f;  // (a lead within the slack -- ties included -- is never followed: the level is evaluated at visit time)
                bestc = b1 ? (int)e1.c : (int)e0.c;
                MZ_TS(5);  // [5] backup pass 2: per-action loop
                SelCache cc;
                cc.packed = (b1 ? 1 : 0) | ((st.epoch & 0xff) << 8) | ((bestc < 0 ? P.NN : bestc) << 16);
                cc.t = b1 ? ud10 : -ud10;
                cc.a = norm ? (b1 ? a10 : -a10) : __uint_as_float(0xff800000u);  // raw-Q levels (before normalisation switches on) are evaluated at visit time
                cc.k = b1 ? k10 : -k10;
                *cache_at(smem, P, e, p) = cc;
                decided = norm & (fabsf(d10) > kCacheSlack);  // == what the next descent's test computes from cc (negation is exact)
            } else {
                if constexpr (ACT > 0) {
                    Entry2 en[ACT];
                    double f[ACT], pr[ACT];
#pragma unroll
                    for (int j = 0; j < ACT; j++) en[j] = er[j];
#pragma unroll
                    for (int j = 0; j < ACT; j++) { f[j] = frow[en[j].cn]; pr[j] = prior[j]; }
                    // the same ranking as rank_actions with two simplifications that change no result: the form of child_U is chosen once
                    // per backup, not per action (two copies of the loop), and ties are not counted -- an action that ties with the best
                    // one becomes the runner-up, margin 0, which the next descent's test (thr < t) can never pass: the level is then
                    // evaluated (and its draw consumed) at visit time exactly as with the explicit tie mark
                    int bestcw = -1;  // {visits, node} word of the best child's entry (node -1: unexpanded)
                    auto refresh = [&](auto f32_tag) {
                        constexpr bool F32 = decltype(f32_tag)::value;
#pragma unroll
                        for (int j = 0; j < ACT; j++) {
                            const float ua = F32 ? ((float)pr[j] * (float)f[j]) : (float)(pr[j] * f[j]);  // child_u
                            er[j].U = ua;
                            const float qn = (float)((en[j].vq - q_sub) * q_mul);
                            const float u = (en[j].cn > 0 ? qn : 0.0f) + ua;
                            // runner-up = the median of {best, runner-up, u} (best >= runner-up throughout): one v_med3_f32 for a compare
                            // and two selects; the best child's node travels as the entry's {visits, node} word and is unpacked once
                            const bool gt = u > best;
                            int cw;
                            __builtin_memcpy(&cw, &en[j].cn, 4);
                            second = __builtin_amdgcn_fmed3f(best, second, u);
                            besta = gt ? j : besta;
                            bestcw = gt ? cw : bestcw;
                            best = gt ? u : best;
                        }
                    };
                    if (prior_f32) refresh(std::true_type{});
                    else refresh(std::false_type{});
                    bestc = bestcw >> 16;
                    cnt = 1;  // (ties: see above)
                } else {
                    rank_actions(std::integral_constant<int, 4>{});
                }
                MZ_TS(5);  // [5] backup pass 2: per-action loop
                SelCache cc;  // real tie: action 0xff = evaluate (and draw) at visit time; single action: +inf margin
                cc.packed = ((cnt == 1) ? besta : 0xff) | ((st.epoch & 0xff) << 8) | ((bestc < 0 ? P.NN : bestc) << 16);
                cc.t = (cnt == 1) ? __double2float_rd(((double)best - (double)second) + 2.0 * st.drift) : __uint_as_float(0xff800000u);
                cc.a = 0.0f; cc.k = 0.0f;
                *cache_at(smem, P, e, p) = cc;
                decided = thr < cc.t;
            }
        }
        if (L <= 16) {
            // lane a0 owns path position L-1-a0 (lane 0: the new leaf, lane L-1: the root); the next descent retraces the
            // path while every node from the root down is decided AND its choice is the next path node (lane a0 - 1)
            const int pnext = dpp_i<DPP_SHR1>(p);
            const bool ok = decided && a0 >= 1 && bestc == pnext;
            const unsigned okm = (unsigned)(__ballot(ok) >> (tid & 48)) & 0xffffu;
            const unsigned stop = (~okm & ((1u << L) - 1u)) | 1u;  // lanes at which the retrace stops; the leaf lane always does
            const int r = 31 - __clz((int)stop);                  // the one nearest to the root
            const int pr = __shfl(p, (tid & 48) | r, 64);
            resume = env_ok ? (pr | ((L - 1 - r) << 16)) : 0;
        }
    }
    MZ_TS(6);  // [6] backup pass 2: cache write + resume point
    MZ_TS_COUNT(9);
    MZ_TS_FLUSH(12);
    return resume;
}

__device__ __forceinline__ void tree2_finish(unsigned char* smem, const SearchParams& P, int e, int env_g) {
    int* rv = reinterpret_cast<int*>(smem + P.t_pi0) + e * P.A;
    const Entry2* er = entry2_row(smem, P, e, 0);
    for (int a = 0; a < P.A; a++) rv[a] = er[a].cn;
    const Node2* root = node2_at(smem, P, e, 0);
    play_from_visits(smem, P, e, env_g, rv, root->W, root->N);
}

// The same play step (generate_play_policy + the action draw, mcts.py:391-407) by the env's 16 lanes, A <= 16: lane a owns
// action a -- its mask byte, its power, its division and its cdf prefix -- where one lane ran ten dependent global loads and
// twenty float64 divisions in a row (TicTacToe: 20 k cycles per move).  The sums keep the reference's orders: numpy's pairwise
// np.sum over the A powers (every lane adds them itself), left-to-right prefix sums for np.cumsum.  Results identical to
// play_from_visits; all 16 lanes of the env must call it.
// `out_action` / `out_root`: the sampled action and the root value, segment-uniform, for a caller that goes on to step the env;
// the policy stays in the env's t_tmp row (float64).
__device__ __forceinline__ void tree2_finish_group(unsigned char* smem, const SearchParams& P, int e, int a0, int env_g, int& out_action, double& out_root) {
    const int A = P.A;
    const bool mine = a0 < A;
    const int a = mine ? a0 : 0;
    double* tmp = reinterpret_cast<double*>(smem + P.t_tmp) + e * A;
    const Node2 root = *node2_at(smem, P, e, 0);
    int v = entry2_row(smem, P, e, 0)[a].cn;
    if (P.has_mask && !P.mask[(size_t)env_g * A + a]) v = 0;
    const bool legal = !P.has_mask || P.mask[(size_t)env_g * A + a] != 0;
    if (mine && P.out_visits) P.out_visits[(size_t)env_g * A + a] = v;
    const double T = P.temperature[env_g];
    double ex = 1.0;
    if (T > 0.0) {
        ex = 1.0 / T;
        ex = ex < 5.0 ? ex : 5.0;
        ex = ex > 1.0 ? ex : 1.0;
    }
    // argmax with the first maximum winning: key = visits * 16 + (15 - action)
    int best = 15 - (row_max_i(mine ? v * 16 + (15 - a) : -1) & 15);
    double t = (T > 0.0) ? pow_policy((double)v, ex) : (double)v;
    if (mine) tmp[a] = t;
    double s = np_sum_f64(tmp, A);  // (LDS operations of a wave are in order: every lane sees all A powers)
    if (!(s > 0.0)) {  // every visit went to an illegal root child: see play_from_visits
        if (P.rng_mode != 0) {
            const unsigned lm = (unsigned)(__ballot(mine && legal) >> (__lane_id() & 48u)) & 0xffffu;
            t = (mine && legal) ? 1.0 : 0.0;
            s = 0.0;
            for (int j = 0; j < A; j++) s = s + (((lm >> j) & 1u) ? 1.0 : 0.0);
            best = lm ? __ffs((int)lm) - 1 : 0;
        }
    }
    const double pi = t / s;
    if (mine) tmp[a] = pi;
    // lane 0 stores the whole policy: the fused env step's record (env_record, same lane) reads it back from global memory
    // moments later, and only a lane's OWN stores are certain to be visible to its loads
    if (a0 == 0)
        for (int j = 0; j < A; j++) P.out_pi[(size_t)env_g * A + j] = tmp[j];
    int action = best;
    if (!P.deterministic) {
        double uu;
        if (P.rng_mode == 0) uu = P.u_final[env_g];
        else {
            Philox g(P.seed, P.env_offset + (unsigned)env_g, P.move_counter, 0x30000000u);
            uu = g.uniform();
            if (P.dbg_ufinal && a0 == 0) P.dbg_ufinal[env_g] = uu;
        }
        // np.random.choice(p=pi): cdf = cumsum(pi); cdf /= cdf[-1]; searchsorted(cdf, u, side='right')
        double c = 0.0, last = 0.0;
        for (int j = 0; j < A; j++) {
            last = last + tmp[j];
            c = j <= a ? last : c;
        }
        const unsigned le = (unsigned)(__ballot(mine && (c / last <= uu)) >> (__lane_id() & 48u)) & 0xffffu;
        const int idx = le ? 32 - __clz((int)le) : 0;  // one past the last action whose cdf value is <= u
        action = idx >= A ? A - 1 : idx;
    }
    const double rootv = root.N > 0 ? root.W / (double)root.N : 0.0;
    if (a0 == 0) {
        P.out_action[env_g] = action;
        P.out_root[env_g] = rootv;
    }
    out_action = action;
    out_root = rootv;
}
__device__ __forceinline__ void tree2_finish_group(unsigned char* smem, const SearchParams& P, int e, int a0, int env_g) {
    int action;
    double rootv;
    tree2_finish_group(smem, P, e, a0, env_g, action, rootv);
}

// root_prior (mz_search.h) by the env's 16 lanes, A <= 16: lane a owns action a's mask byte, mix and divisions; the sums are
// added by every lane itself in the reference's orders (the production noise's plain left-to-right sum, numpy's np.sum for the
// renormalisation).  Same values as root_prior; all 16 lanes of the env must call it.
__device__ __forceinline__ void root_prior_group(unsigned char* smem, const SearchParams& P, int e, int a0, int env_g) {
    const int A = P.A;
    const bool mine = a0 < A;
    const int a = mine ? a0 : 0;
    double* prior = reinterpret_cast<double*>(smem + P.t_prior) + e * A;
    float* pi0 = reinterpret_cast<float*>(smem + P.t_pi0) + e * A;
    double* tmp = reinterpret_cast<double*>(smem + P.t_tmp) + e * A;
    const bool legal = !P.has_mask || P.mask[(size_t)env_g * A + a] != 0;
    if (P.noise_mode != 0) {
        double nz;
        if (P.noise_mode == 1) {
            nz = P.noise[(size_t)env_g * A + a];
        } else {  // tmp[] holds the gamma draws of root_noise_lanes
            double s = 0.0;
            for (int j = 0; j < A; j++) s += tmp[j];
            nz = s > 0.0 ? tmp[a] / s : 1.0 / (double)A;
            if (P.dbg_noise && mine) P.dbg_noise[(size_t)env_g * A + a] = nz;
        }
        const float om = (float)(1.0 - P.eps);
        const float t = om * pi0[a];      // float32 product (python scalar * float32 array)
        const double en = P.eps * nz;     // float64
        double pr = (double)t + en;
        if (P.has_mask) {
            if (!legal) pr = 0.0;
            if (mine) prior[a] = pr;
            const double s = np_sum_f64(prior, A);  // (LDS operations of a wave are in order)
            if (s > 0) pr = pr / s;
        }
        if (mine) prior[a] = pr;
    } else {
        float p0 = pi0[a];
        if (P.has_mask) {
            if (!legal) p0 = 0.0f;
            if (mine) pi0[a] = p0;
            const float s = np_sum_f32(pi0, A);
            if (s > 0) p0 = p0 / s;
            if (mine) pi0[a] = p0;
        }
        if (mine) prior[a] = (double)p0;
    }
}

