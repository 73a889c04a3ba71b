// mz_search_fast.h -- the tuned search kernel for the benchmark shapes (hidden_dim 64, num_planes 256 or 512,
// value/reward support <= 32, <= 16 actions: CartPole, LunarLander, TicTacToe MLP nets).  Same algorithm, same LDS
// tree, same numerics as k_search (mz_search.h); what changes is how the per-simulation network evaluation is fed:
//
//   * "chain" layers (K = num_planes: dynamics layer 2, reward layer 2, value layer 2) have one 16-neuron tile per
//     wave and a 4*P/16-deep dependent MFMA chain.  Their weight tile streams through an 8-group register ring that
//     is primed one phase early (during the preceding wide layer) and refilled 8 groups (1280 chain cycles) ahead.
//   * "wide" layers (N = num_planes: dynamics layer 1, reward layer 1, value layer 1) stream their weights from L2
//     through a double-buffered register ring, in a per-wave contiguous "stream" layout, one group (8 tiles x 16 k)
//     ahead of the MFMAs -- including across layer and simulation boundaries (the next layer's first group is
//     requested before the current layer's epilogue / barriers / tree phase), so the loads are never waited for.
//
// Stream layout (host packs it, see pack_stream in planner.hip): for wave w, layer with NT tiles per wave
// (tile t = w + 4 j) and KG k-groups, block (g, j) is the float4[64] at  ((w * KG + g) * NT + j) * 64 + lane.
#pragma once
#include <type_traits>

#include "mz_search.h"

namespace mz {

struct FastWeights {
    const float4* dyn0;  // stream, NT = P/64, KG = ceil((64+A)/16)
    const float4* rew0;  // stream, NT = P/64, KG = 4
    const float4* val0;  // stream, NT = P/64, KG = 4
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Weight loads go through buffer descriptors (SRD in SGPRs): address = base + voffset (lane * 16, one VGPR) + soffset
// (wave-uniform byte offset in an SGPR).  With flat global loads hipcc materialises one 64-bit VGPR address per
// unrolled load and keeps hundreds of them live across the simulation loop (spills); here no address VGPRs exist.
struct WSrc {
    __amdgpu_buffer_rsrc_t r;
    int base;  // wave-uniform byte offset of this wave's first block
};

__device__ __forceinline__ WSrc make_wsrc(const void* p, unsigned bytes, int base) {
    WSrc w;
    w.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
    w.base = base;
    return w;
}

__device__ __forceinline__ float4 bload(const WSrc& w, int voff, int block) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w.r, voff, w.base + block * 1024, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <int NT>
__device__ __forceinline__ void wload(float4 (&w)[NT], const WSrc& src, int voff, int block0) {
#pragma unroll
    for (int j = 0; j < NT; j++) w[j] = bload(src, voff, block0 + j);
}

template <int NT>
__device__ __forceinline__ void wmma4(f32x4 (&acc)[NT], const float4 (&w)[NT], const float4 x) {
#pragma unroll
    for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].x, x.x, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].y, x.y, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].z, x.z, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].w, x.w, acc[j], 0, 0, 0);
}

template <int NT>
__device__ __forceinline__ void wmma_rem(f32x4 (&acc)[NT], const float4 (&w)[NT], const float4 x, int rem) {
#pragma unroll
    for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].x, x.x, acc[j], 0, 0, 0);
    if (rem > 1) {
#pragma unroll
        for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].y, x.y, acc[j], 0, 0, 0);
    }
    if (rem > 2) {
#pragma unroll
        for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].z, x.z, acc[j], 0, 0, 0);
    }
    if (rem > 3) {
#pragma unroll
        for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].w, x.w, acc[j], 0, 0, 0);
    }
}

// One streamed wide layer for this wave.  On entry buf[PAR] holds group 0 of this layer (requested earlier); on exit
// buf[(PAR + KG) & 1] holds group 0 of the NEXT layer (`next`), requested while the last group was being multiplied.
// `last_steps` = k-steps in the last group (1..4).
template <int NT, int KG, int PAR, typename Epi>
__device__ __forceinline__ void stream_layer(float4 (&buf)[2][NT], const WSrc& cur, const WSrc& next, const float* __restrict__ bias,
                                             const float* __restrict__ Xs, int last_steps, int wave, int lane, Epi epi) {
    const int voff = lane * 16;
    const int q = lane >> 4;
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + (wave + 4 * j) * 16 + q * 4);
        acc[j] = f32x4{bv.x, bv.y, bv.z, bv.w};
    }
    const float4* xp = reinterpret_cast<const float4*>(Xs) + lane;
#pragma unroll
    for (int g = 0; g < KG; g++) {
        if (g + 1 < KG) wload<NT>(buf[(PAR + g + 1) & 1], cur, voff, (g + 1) * NT);
        else wload<NT>(buf[(PAR + g + 1) & 1], next, voff, 0);
        const float4 x = xp[g * 64];
        if (g + 1 < KG) wmma4<NT>(acc, buf[(PAR + g) & 1], x);
        else wmma_rem<NT>(acc, buf[(PAR + g) & 1], x, last_steps);
        // issue order inside the group: the B-operand read, then one weight load per 4 MFMAs (the loads' issue slots hide
        // under the 32-cycle MFMAs instead of preceding them)
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int i = 0; i < NT; i++) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the ring 1 group deep: no hoisting of later groups' loads
    }
#pragma unroll
    for (int j = 0; j < NT; j++) epi(wave + 4 * j, acc[j]);
}

// One chain layer tile with register-resident weights: 4*KGP dependent MFMAs, B operand from LDS.
template <int KGP, typename Epi>
__device__ __forceinline__ void chain_layer(const float4 (&wres)[KGP], const float* __restrict__ bias, int tile, const float* __restrict__ Xs,
                                            int lane, Epi epi) {
    const int q = lane >> 4;
    const float4 bv = *reinterpret_cast<const float4*>(bias + tile * 16 + q * 4);
    f32x4 acc = f32x4{bv.x, bv.y, bv.z, bv.w};
    const float4* xp = reinterpret_cast<const float4*>(Xs) + lane;
    float4 x = xp[0], x1 = xp[64];
#pragma unroll
    for (int g = 0; g < KGP; g++) {
        const float4 x2 = xp[(g + 2 < KGP ? g + 2 : g) * 64];  // B operand two groups ahead of the chain
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[g].x, x.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[g].y, x.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[g].z, x.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[g].w, x.w, acc, 0, 0, 0);
        x = x1;
        x1 = x2;
        __builtin_amdgcn_sched_barrier(0);
    }
    epi(tile, acc);
}

// Chain layer tile with its weights streamed through a D-deep register ring: on entry ring[d] holds group d
// (requested by chain_prime well before); every consumed slot is immediately re-requested D groups ahead, so D * 4
// dependent MFMAs (D * 160 cycles) of latency cover is always in flight.
template <int KGP, int D>
__device__ __forceinline__ void chain_prime(float4 (&ring)[D], const WSrc& wp, int lane) {
#pragma unroll
    for (int d = 0; d < D; d++) ring[d] = bload(wp, lane * 16, d);
}

template <int KGP, int D>
__device__ __forceinline__ f32x4 chain_layer_ring(float4 (&ring)[D], const WSrc& wp, const float* __restrict__ bias, int tile,
                                                  const float* __restrict__ Xs, int lane) {
    const int q = lane >> 4;
    const float4 bv = *reinterpret_cast<const float4*>(bias + tile * 16 + q * 4);
    f32x4 acc = f32x4{bv.x, bv.y, bv.z, bv.w};
    const float4* xp = reinterpret_cast<const float4*>(Xs) + lane;
    float4 x = xp[0], x1 = xp[64];
#pragma unroll
    for (int g = 0; g < KGP; g++) {
        const float4 x2 = xp[(g + 2 < KGP ? g + 2 : g) * 64];
        const float4 w = ring[g % D];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x.w, acc, 0, 0, 0);
        if (g + D < KGP) ring[g % D] = bload(wp, lane * 16, g + D);
        x = x1;
        x1 = x2;
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// P = num_planes (256 or 512); XG = k-groups of the dynamics input (hidden 64 + one-hot A): 5 for A <= 16
// FUSE: device self-play with the environment inside this kernel (mz_selfplay_step on short moves)
template <int P, bool FUSE = false>
__global__ __launch_bounds__(WG_THREADS) void k_search_fast(const SearchParams Pm, const FastWeights FW) {
    constexpr int NT = P / 64;   // wide-layer tiles per wave
    constexpr int KGP = P / 16;  // k-groups of the chain layers
    constexpr int XG = 5;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, e = tid >> 4, a0 = tid & 15, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int env_g = blockIdx.x * TILE_E + e;
    const bool env_ok = env_g < Pm.B;
    if constexpr (FUSE) fused_env_pre(Pm, a0, env_g, env_ok);
    const MlpNet& net = Pm.net;
    const MlpLds& o = Pm.o;
    float* pi0 = reinterpret_cast<float*>(smem + Pm.t_pi0);
    const float** src = reinterpret_cast<const float**>(smem + Pm.t_ptr);
    float** dst = reinterpret_cast<float**>(smem + Pm.t_ptr) + 16;

    MZ_STAMP_DECL
    MZ_STAMP_START();
    // ---- tables, tree, root (identical to k_search) ----
    stage_biases(net, lds, tid);
    {
        if (Pm.tree_mode == 2) {
            tree2_init(smem, Pm, tid, env_ok, env_g);
        } else {
            double* ft = reinterpret_cast<double*>(smem + Pm.t_ftab);
            for (int i = tid; i < (Pm.S + 1) * (Pm.S + 1); i += WG_THREADS) ft[i] = Pm.ftab[i];
            short* ch = reinterpret_cast<short*>(smem + Pm.t_child);
            for (int i = tid; i < TILE_E * Pm.NN * Pm.A; i += WG_THREADS) ch[i] = -1;
            if (a0 == 0) {
                TreeNode* r = node_at(smem, Pm, e, 0);
                r->W = 0.0; r->vq = 0.0; r->N = 0; r->reward = 0.0f; r->parent = -1; r->move = -1;
                r->player = env_ok ? Pm.cur[env_g] : 0;
            }
        }
        if (a0 == 0) {
            double* mm = reinterpret_cast<double*>(smem + Pm.t_mm) + e * 2;
            mm[0] = Pm.has_bounds ? Pm.kb_min : __longlong_as_double(0x7ff0000000000000LL);
            mm[1] = Pm.has_bounds ? Pm.kb_max : __longlong_as_double(0xfff0000000000000LL);
            int* sel = reinterpret_cast<int*>(smem + Pm.t_sel) + e * 4;
            sel[0] = sel[1] = sel[2] = sel[3] = 0;
            src[e] = env_ok ? Pm.obs + (size_t)env_g * net.in_dim : nullptr;
            dst[e] = env_ok ? Pm.hidden + (size_t)env_g * Pm.NN * net.H : nullptr;
        }
    }
    __syncthreads();
    root_noise_lanes(smem, Pm, e, a0, env_g, env_ok);
    load_obs(net, lds + o.X, src, tid);
    __syncthreads();
    mlp_initial_tile(net, o, lds, dst, pi0, tid);
    __syncthreads();
    if (a0 == 0 && env_ok) root_prior(smem, Pm, e, env_g);

    // ---- chain-layer weight sources: dynamics layer 2 tile `wave`; reward layer 2 (waves 0,1) / value layer 2 (waves 2,3) ----
    constexpr int HD = 8;  // chain ring depth (groups in flight): shared by the dynamics-2 chain and the head chain
    float4 cring[HD];
    const WSrc pd = make_wsrc(net.L[L_DYN1].w, (unsigned)net.L[L_DYN1].n_tiles * KGP * 1024u, wave * KGP * 1024);
    const bool is_val_wave = wave >= 2;
    const int head_tile = wave & 1;
    const MlpLayer& HL = is_val_wave ? net.L[L_VAL1] : net.L[L_REW1];
    const bool head_ok = head_tile < HL.n_tiles;
    const WSrc ph = make_wsrc(HL.w, (unsigned)HL.n_tiles * KGP * 1024u, (head_ok ? head_tile : 0) * KGP * 1024);
    // ---- streamed wide-layer weights: per-wave stream bases; ring primed with dynamics layer 1, group 0 ----
    const WSrc s_dyn0 = make_wsrc(FW.dyn0, 4u * XG * NT * 1024u, wave * XG * NT * 1024);
    const WSrc s_rew0 = make_wsrc(FW.rew0, 4u * 4 * NT * 1024u, wave * 4 * NT * 1024);
    const WSrc s_val0 = make_wsrc(FW.val0, 4u * 4 * NT * 1024u, wave * 4 * NT * 1024);
    float4 ring[2][NT];
    wload<NT>(ring[0], s_dyn0, lane * 16, 0);
    const int x_last = net.L[L_DYN0].k_steps - 4 * (XG - 1);  // k-steps in the last input group (1..4)
    // MFMA-side env of this lane (D column) and its hidden-state rows in the HBM node store
    const int e2 = lane & 15, q = lane >> 4;
    const int env2 = blockIdx.x * TILE_E + e2;
    const bool env2_ok = env2 < Pm.B;
    float* const hid_sel = Pm.hidden + (size_t)(env_ok ? env_g : 0) * Pm.NN * 64;   // select-side env (tid >> 4)
    float* const hid_mma = Pm.hidden + (size_t)(env2_ok ? env2 : 0) * Pm.NN * 64;  // MFMA-side env (lane & 15)
    const float* bias = lds;  // biases live in LDS (stage_biases)
    __syncthreads();
    MZ_STAMP(0);  // root: tables + initial inference + prior

    // Ring parity: a layer that starts with its group 0 in ring[PAR] and has KG groups leaves its successor's group 0
    // in ring[(PAR + KG) & 1].  dyn0 has 5 groups, rew0 and val0 4 each, so the parity flips once per simulation:
    // the body is instantiated for both parities and the simulation loop alternates them (no register copies).
    auto sim = [&](auto par_tag, int s) {
        constexpr int PAR = decltype(par_tag)::value;
        int lp, la, mypath = 0;
        if (Pm.tree_mode == 2) tree2_select(smem, Pm, tid, env_ok, env_g, lp, la, mypath);
        else tree_select(smem, Pm, tid, env_ok, env_g, lp, la);
        // gather: the env's 16 lanes fetch the parent's hidden state (64 floats = one float4 per lane) straight after
        // their descent and scatter it, with the one-hot action (network.py:191-193), into the packed B-operand buffer
        {
            float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (env_ok) hv = *reinterpret_cast<const float4*>(hid_sel + (size_t)lp * 64 + a0 * 4);
            float* X = lds + o.X;
            const int base = ((((a0 >> 2) * 64) + e) << 2) + (a0 & 3);  // pk(4*a0 + j, e) = base + 64 j
            X[base] = hv.x; X[base + 64] = hv.y; X[base + 128] = hv.z; X[base + 192] = hv.w;
            X[(((4 * 64) + (a0 & 3) * 16 + e) << 2) + (a0 >> 2)] = (a0 == la && a0 < Pm.A) ? 1.0f : 0.0f;  // pk(64 + a0, e)
        }
        chain_prime<KGP, HD>(cring, pd, lane);  // dynamics-2 chain weights: first HD groups land during dynamics layer 1
        __syncthreads();
        MZ_STAMP(1);  // select + gather
        // dynamics layer 1 (wide, streamed): X -> H1
        stream_layer<NT, XG, PAR>(ring, s_dyn0, s_rew0, bias + net.L[L_DYN0].b_lds, lds + o.X, x_last, wave, lane, EpiReluPacked{lds + o.H1, lane});
        __syncthreads();
        MZ_STAMP(3);  // dynamics layer 1
        // dynamics layer 2 (chain): H1 -> h; normalisation (util.py:31-36) fused into the epilogue: per-wave min/max
        // partials through LDS, then every lane normalises its own 4 neurons in registers
        {
            const f32x4 h = chain_layer_ring<KGP, HD>(cring, pd, bias + net.L[L_DYN1].b_lds, wave, lds + o.H1, lane);
            float mn = h[0] < h[1] ? h[0] : h[1], mx = h[0] > h[1] ? h[0] : h[1];
            mn = h[2] < mn ? h[2] : mn; mx = h[2] > mx ? h[2] : mx;
            mn = h[3] < mn ? h[3] : mn; mx = h[3] > mx ? h[3] : mx;
            float t;
            t = __shfl_xor(mn, 16, 64); mn = t < mn ? t : mn;
            t = __shfl_xor(mx, 16, 64); mx = t > mx ? t : mx;
            t = __shfl_xor(mn, 32, 64); mn = t < mn ? t : mn;
            t = __shfl_xor(mx, 32, 64); mx = t > mx ? t : mx;
            float* pm = lds + o.PM;
            if (q == 0) { pm[wave * 16 + e2] = mn; pm[64 + wave * 16 + e2] = mx; }
            EpiRawPacked{lds + o.HN, lane}(wave, h);  // un-normalised state feeds the reward head (network.py:195-196)
            chain_prime<KGP, HD>(cring, ph, lane);    // head-chain weights: land during the two wide layers below
            __syncthreads();
            MZ_STAMP(4);  // dynamics layer 2 (chain)
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const float a = pm[w * 16 + e2], b = pm[64 + w * 16 + e2];
                mn = a < mn ? a : mn;
                mx = b > mx ? b : mx;
            }
            const float d = (mx - mn) + 1e-8f;
            const f32x4 hs = f32x4{(h[0] - mn) / d, (h[1] - mn) / d, (h[2] - mn) / d, (h[3] - mn) / d};
            EpiRawPacked{lds + o.HS, lane}(wave, hs);
            if (env2_ok) *reinterpret_cast<float4*>(hid_mma + (size_t)(s + 1) * 64 + wave * 16 + q * 4) = make_float4(hs[0], hs[1], hs[2], hs[3]);
        }
        __syncthreads();
        MZ_STAMP(5);  // normalise + hidden store
        // reward layer 1 (HN -> H1) and value layer 1 (HS -> V1), wide, streamed; the last one requests the next
        // simulation's dynamics group 0, which then lands during the chain / softmax / tree phases
        stream_layer<NT, 4, PAR ^ 1>(ring, s_rew0, s_val0, bias + net.L[L_REW0].b_lds, lds + o.HN, 4, wave, lane, EpiReluPacked{lds + o.H1, lane});
        stream_layer<NT, 4, PAR ^ 1>(ring, s_val0, s_dyn0, bias + net.L[L_VAL0].b_lds, lds + o.HS, 4, wave, lane, EpiReluPacked{lds + o.V1, lane});
        __syncthreads();
        MZ_STAMP(6);  // reward + value layer 1
        // reward layer 2 (waves 0,1) / value layer 2 (waves 2,3): chain, weights through the ring
        if (head_ok) {
            if (is_val_wave) {
                const f32x4 lg = chain_layer_ring<KGP, HD>(cring, ph, bias + net.L[L_VAL1].b_lds, head_tile, lds + o.V1, lane);
                EpiLogits{lds + o.LG + 16 * o.lg_stride, o.lg_stride, lane}(head_tile, lg);
            } else {
                const f32x4 lg = chain_layer_ring<KGP, HD>(cring, ph, bias + net.L[L_REW1].b_lds, head_tile, lds + o.H1, lane);
                EpiLogits{lds + o.LG, o.lg_stride, lane}(head_tile, lg);
            }
        }
        __syncthreads();
        MZ_STAMP(7);  // reward / value layer 2 (chain)
        // softmax -> expectation -> signed_parabolic (util.py:70-93) in registers: 16 lanes per row, 2 logits per lane;
        // the results are segment-uniform, so the env's lane 0 goes straight on to expand + backup
        {
            const float* rr = lds + o.LG + e * o.lg_stride;
            const float* rv = lds + o.LG + (16 + e) * o.lg_stride;
            const bool r0 = a0 < net.Sr, r1 = a0 + 16 < net.Sr, v0 = a0 < net.Sv, v1 = a0 + 16 < net.Sv;
            const float lr0 = rr[r0 ? a0 : 0], lr1 = rr[r1 ? a0 + 16 : 0], lv0 = rv[v0 ? a0 : 0], lv1 = rv[v1 ? a0 + 16 : 0];
            const float rew = net.Sr == 1 ? rr[0] : row2_logits_to_scalar(lr0, lr1, r0, r1, net.Sr, a0);
            const float val = net.Sv == 1 ? rv[0] : row2_logits_to_scalar(lv0, lv1, v0, v1, net.Sv, a0);
            MZ_STAMP(8);  // softmax + expectation + transform
            if (Pm.tree_mode == 2) tree2_backup(smem, Pm, tid, env_ok, s, rew, val, mypath);
            else if (a0 == 0 && env_ok) tree_expand_backup(smem, Pm, e, s, rew, val);
        }
        __syncthreads();
        MZ_STAMP(9);  // expand + backup
    };
    int s = 0;
    for (; s + 1 < Pm.S; s += 2) {
        sim(std::integral_constant<int, 0>{}, s);
        sim(std::integral_constant<int, 1>{}, s + 1);
    }
    if (s < Pm.S) sim(std::integral_constant<int, 0>{}, s);
    if (a0 == 0 && env_ok) {
        if (Pm.tree_mode == 2) tree2_finish(smem, Pm, e, env_g);
        else tree_finish(smem, Pm, e, env_g);
        if constexpr (FUSE) env_step_one(Pm.fenv, env_g);  // env.step + record + auto-reset with the action this lane just sampled
    }
    MZ_STAMP(10);  // play policy + action
    MZ_STAMP_FLUSH(Pm);
}

}  // namespace mz
