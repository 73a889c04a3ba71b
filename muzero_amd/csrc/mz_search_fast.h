// mz_search_fast.h -- the tuned search kernel for the benchmark shapes (hidden_dim 64, num_planes 256 or 512,
// value/reward support <= 32, <= 16 actions: CartPole, LunarLander, TicTacToe MLP nets).  Same algorithm, same LDS
// tree, same numerics as k_search (mz_search.h); what changes is how the per-simulation network evaluation runs:
//
//   * REGISTER-RESIDENT ACTIVATIONS.  Wave w computes the contiguous quarter [NT*w, NT*w + NT) of every num_planes-wide
//     hidden layer (NT = P/64 tiles of 16 neurons) and keeps it in its accumulators: a D-layout accumulator IS the B
//     operand of the next layer's MFMAs in the summation order of mz_mlp.h (k-step i of block g = register i of tile g).
//     The following K = num_planes layer is K-split: each wave multiplies its own quarter and the four partial tiles
//     meet in LDS -- ((c0 + c1) + c2) + c3, the order the oracle restates.  The 512 x 16 hidden layers never touch LDS
//     (v1 of this kernel wrote and re-read 32 KiB per layer and barriered after every layer: 7 barriers per
//     simulation; this one has 4).
//   * ONE WEIGHT STREAM PER WAVE.  All six layers' A operands of a simulation are packed on the host in exactly the
//     order wave w consumes them, in slots of NT float4[64] blocks (4 NT MFMAs each); the wave walks the stream
//     cyclically through an RD-deep register ring, RD - 1 slots (~2 k cycles) ahead of the MFMAs -- across layer,
//     barrier and simulation boundaries, so a load is never waited for.  buffer_load with an SGPR offset: no address
//     VGPRs.
//
// Stream layout (host: pack_fast_stream in planner.hip): float4 index ((w * SL + slot) * NT + j) * 64 + lane, slots of
// one simulation in order
//     D1  dynamics layer 1, 5 slots: input block g (4 hidden blocks + the action block), j = tile NT w + j
//     D2  dynamics layer 2 (K-split), 4 slots: j <-> (kb = (slot NT + j) / 4, t = (slot NT + j) % 4): out tile t, input block NT w + kb
//     R1  reward layer 1, 4 slots (input block g of the un-normalised state, network.py:195-196)
//     R2  reward layer 2 (K-split), TR slots: j <-> (kb, t) with TR out tiles
//     V1  value layer 1, 4 slots (normalised state);  V2  value layer 2 (K-split), TV slots
//     zero padding up to a multiple of RD slots.
#pragma once
#include <type_traits>

#include "mz_search.h"

namespace mz {

#ifndef MZ_FAST_RD
#define MZ_FAST_RD 3
#endif
#ifndef MZ_FAST_RD256
#define MZ_FAST_RD256 4
#endif
// depth of the weight ring: the stream runs RD - 1 slots ahead of the MFMAs (a slot is 4 NT MFMAs = 128 NT cycles).  The loads are
// not late at any depth tried (num_planes 512: 3 / 4 / 6 within 0.3 %); at num_planes 256 the ten-action build's 16 slots per
// simulation divide by 4 -- no padding slots -- and 4 deep measures 0.4 % under 3 deep (2 deep: +1.2 %)
// The builds that do not know "two actions" keep more of the tree phases' state live across the MFMA phases: at num_planes 512 a ring
// of 3 x 8 float4 leaves them 6-10 spilled VGPRs (scratch round trips inside the simulation loop); 2 deep they fit.
#ifndef MZ_FAST_RD_GEN
#define MZ_FAST_RD_GEN 2
#endif
constexpr int fast_rd(int planes, int ac) { return planes == 256 ? MZ_FAST_RD256 : (ac == 2 ? MZ_FAST_RD : MZ_FAST_RD_GEN); }
#ifndef MZ_FAST_HW
#define MZ_FAST_HW 1
#endif
constexpr bool kFastHW = MZ_FAST_HW != 0;  // helper waves (see k_search_fast); 0: the 4-wave kernel (A/B measurements)
#ifndef MZ_FAST_SC
#define MZ_FAST_SC 1
#endif
#ifndef MZ_FAST_AX
#define MZ_FAST_AX 1
#endif
// the ten-action instantiation (TicTacToe MLP net: MSE heads, reward / value support 1):
//   SC  the two scalar heads' second layers (num_planes -> 1) as VALU chains instead of one 16-row MFMA tile with 15 idle rows each
//   AX  the dynamics net's one-hot action block as ONE addition of the action's weight column instead of 3 k-steps of MFMAs
constexpr bool kFastSC = MZ_FAST_SC != 0, kFastAX = MZ_FAST_AX != 0;

struct FastWeights {
    const float4* stream;
    unsigned bytes;  // whole stream (4 waves)
    const float* wact;  // AX: the dynamics net's action columns, [A][num_planes] (row-contiguous per action), or null
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Weight loads go through buffer descriptors (SRD in SGPRs): address = base + voffset (lane * 16, one VGPR) + soffset
// (wave-uniform byte offset in an SGPR).  With flat global loads hipcc materialises one 64-bit VGPR address per
// unrolled load and keeps hundreds of them live across the simulation loop (spills); here no address VGPRs exist.
struct WSrc {
    __amdgpu_buffer_rsrc_t r;
    int base;  // wave-uniform byte offset of this wave's first block
    int cur;   // running byte offset of the next slot to request (one SGPR, advanced by one s_add per slot)
};

__device__ __forceinline__ WSrc make_wsrc(const void* p, unsigned bytes, int base) {
    WSrc w;
    w.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
    w.base = base;
    w.cur = base;
    return w;
}

__device__ __forceinline__ float4 bload(const WSrc& w, int voff, int block) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w.r, voff, w.base + block * 1024, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ float comp(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// P = num_planes; TR / TV = 16-neuron tiles of the reward / value support; RD = ring depth
template <int P, int TR, int TV, int RD, int XG_ = 5>
struct FastCfg {
    static constexpr int NT = P / 64;
    static constexpr int XG = XG_;  // input blocks of the dynamics net: hidden 64 + one action block (A <= 16); AX: 4, no action block
    static constexpr int I_D1 = 0, I_D2 = I_D1 + XG, I_R1 = I_D2 + 4, I_R2 = I_R1 + 4, I_V1 = I_R2 + TR, I_V2 = I_V1 + 4, I_END = I_V2 + TV;
    static constexpr int SL = (I_END + RD - 1) / RD * RD;  // slots per simulation incl. padding
};

// request slot I + RD - 1 (cyclic) into the ring entry that slot I - 1 has just vacated
template <typename C, int RD, int I>
__device__ __forceinline__ void prefetch_slot(float4 (&ring)[RD][C::NT], WSrc& ws, int voff) {
    constexpr int T = (I + RD - 1) % C::SL;
#ifdef MZ_EXP_NOLOAD
    return;
#endif
    if constexpr (T < C::I_END) {  // padding slots hold nothing: only the ring position advances
        // The slot's byte offset is made opaque here: left to itself hipcc treats the 168 (slot, tile) offsets of a simulation
        // as loop invariants, runs out of SGPRs, parks them in VGPR lanes and then pays v_readlane + 5 wait states in front of
        // EVERY load -- 5 cycles per MFMA over the whole network.  One s_add per slot and the 12-bit immediate field instead.
        int so = ws.cur;  // == ws.base + T * NT * 1024
        asm volatile("" : "+s"(so));
        ws.cur = so + (T + 1 == C::I_END ? -(C::I_END - 1) : 1) * C::NT * 1024;  // the next real slot (the stream is cyclic)
#pragma unroll
        for (int j = 0; j < C::NT; j++) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ws.r, voff + (j & 3) * 1024, so + (j >> 2) * 4096, 0);
            ring[T % RD][j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    }
}

// issue order inside a slot: (the B-operand LDS read,) then one weight load per 4 MFMAs -- the loads' issue slots hide
// under the 32-cycle MFMAs instead of preceding them
// NV > 0: NV VALU instructions of independent work (the reward row's softmax, computed while the value head multiplies)
// ride in every MFMA's shadow: an MFMA holds the issue port for 8 of its 32 cycles, the rest is free for them
template <int NT, bool LDS_READ, int NV = 0>
__device__ __forceinline__ void slot_schedule() {
    if (LDS_READ) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int i = 0; i < NT; i++) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (NV == 0) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        } else {
#pragma unroll
            for (int m = 0; m < 2; m++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
            }
        }
    }
    if (NV == 0) __builtin_amdgcn_sched_barrier(0);  // keep the ring RD - 1 slots deep: no hoisting of later slots' loads
    else __builtin_amdgcn_sched_barrier(0);
}

// The slot loops need the slot index as a compile-time constant for the ring entry; `#pragma unroll` loops over a
// constexpr bound give that after unrolling, but the prefetch helper takes it as a template argument, so the loops are
// written with an index_sequence-style recursion.
// side work placed one stage per slot (between two scheduling fences, so it can only ride in that slot's MFMA shadows)
struct NoHook {
    template <int K>
    __device__ __forceinline__ void stage() {}
};

template <typename C, int RD, int I0, int G, int KG, bool FROM_LDS, int NV = 0, typename Hook = NoHook, int HK0 = 0>
struct WideSlots {
    // FROM_LDS: B operand = float4 read from the packed LDS buffer; else B operand = hin[G] (D-layout registers)
    static __device__ __forceinline__ void run(float4 (&ring)[RD][C::NT], WSrc& ws, int voff, const float4 (&xs)[5], const f32x4 (&hin)[4],
                                               int last_steps, f32x4 (&acc)[C::NT], Hook& hook) {
        constexpr int NT = C::NT, I = I0 + G;
        prefetch_slot<C, RD, I>(ring, ws, voff);
        hook.template stage<HK0 + G>();
        float b[4];
        if (FROM_LDS) {  // all of the layer's B operands were read from LDS before its first MFMA (xs)
            const float4 x = xs[G];
            b[0] = x.x; b[1] = x.y; b[2] = x.z; b[3] = x.w;
        } else {
            b[0] = hin[G][0]; b[1] = hin[G][1]; b[2] = hin[G][2]; b[3] = hin[G][3];
        }
        const float4(&w)[NT] = ring[I % RD];
#pragma unroll
        for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].x, b[0], acc[j], 0, 0, 0);
        if (G + 1 < KG || last_steps > 1) {
#pragma unroll
            for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].y, b[1], acc[j], 0, 0, 0);
        }
        if (G + 1 < KG || last_steps > 2) {
#pragma unroll
            for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].z, b[2], acc[j], 0, 0, 0);
        }
        if (G + 1 < KG || last_steps > 3) {
#pragma unroll
            for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].w, b[3], acc[j], 0, 0, 0);
        }
        slot_schedule<NT, false, NV>();
        if constexpr (G + 1 < KG) WideSlots<C, RD, I0, G + 1, KG, FROM_LDS, NV, Hook, HK0>::run(ring, ws, voff, xs, hin, last_steps, acc, hook);
    }
};

// K-split layer, this wave's quarter: TO output tiles, input = the NT tiles this wave holds in registers (hin).
// Slot D holds NT float4: entry jj <-> (kb = (D NT + jj) / TO, t = (D NT + jj) % TO).
template <typename C, int RD, int I0, int D, int TO, int NV = 0, typename Hook = NoHook, int HK0 = 0>
struct SplitSlots {
    static __device__ __forceinline__ void run(float4 (&ring)[RD][C::NT], WSrc& ws, int voff, const f32x4 (&hin)[C::NT], f32x4 (&acc)[TO],
                                               Hook& hook) {
        constexpr int NT = C::NT, I = I0 + D, KPS = NT / TO;
        static_assert(NT % TO == 0, "tiles per slot");
        prefetch_slot<C, RD, I>(ring, ws, voff);
        hook.template stage<HK0 + D>();
        const float4(&w)[NT] = ring[I % RD];
#pragma unroll
        for (int kk = 0; kk < KPS; kk++) {
            const int kb = D * KPS + kk;
#pragma unroll
            for (int st = 0; st < 4; st++) {
#pragma unroll
                for (int t = 0; t < TO; t++)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(w[kk * TO + t], st), hin[kb][st], acc[t], 0, 0, 0);
            }
        }
        slot_schedule<NT, false, NV>();
        if constexpr (D + 1 < TO) SplitSlots<C, RD, I0, D + 1, TO, NV, Hook, HK0>::run(ring, ws, voff, hin, acc, hook);
    }
};

// padding slots: nothing to multiply, but the ring must keep turning (their prefetches are the next simulation's first slots)
template <typename C, int RD, int I>
struct PadSlots {
    static __device__ __forceinline__ void run(float4 (&ring)[RD][C::NT], WSrc& ws, int voff) {
        if constexpr (I < C::SL) {
            prefetch_slot<C, RD, I>(ring, ws, voff);
            __builtin_amdgcn_sched_barrier(0);
            PadSlots<C, RD, I + 1>::run(ring, ws, voff);
        }
    }
};

// min / max of two non-NaN floats as ONE v_med3_f32 (fminf / fmaxf add a canonicalising v_max x, x per operand)
// one v_med3_f32 each: with a literal infinity hipcc folds fmed3 into maxnum / minnum and renders those as canonicalise
// (v_max x, x) + the operation -- two issue slots per element on ~170 elements per simulation -- so the infinity is made opaque.
// (Inline-asm v_max / v_min would be shorter still, but the hazard recogniser does not see into asm: no wait states after the
// MFMA that produces the operand -- measured as a parity failure.)  Operands are never NaN here.
__device__ __forceinline__ float opaque_f32(unsigned int bits) {
    asm("" : "+s"(bits));
    return __uint_as_float(bits);
}
__device__ __forceinline__ float fmin2(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, opaque_f32(0xff800000u)); }
__device__ __forceinline__ float fmax2(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, opaque_f32(0x7f800000u)); }
// all-lane min / max over the four 16-lane rows of a wave (same lane & 15): v_permlane16_swap / v_permlane32_swap
// exchange whole rows between two registers, so min(r[0], r[1]) is the xor-16 (xor-32) butterfly step in every lane
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rows_min(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmin2(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmin2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float rows_max(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int N>
__device__ __forceinline__ void relu_tiles(f32x4 (&a)[N]) {
    const float inf = opaque_f32(0x7f800000u);
#pragma unroll
    for (int j = 0; j < N; j++) {
#pragma unroll
        for (int r = 0; r < 4; r++) a[j][r] = __builtin_amdgcn_fmed3f(a[j][r], 0.0f, inf);
    }
}

// logit n of env e from the four waves' partial tiles of a K-split head: ((c0 + c1) + c2) + c3
template <int TO>
__device__ __forceinline__ float head_logit(const float* PB, int n, int e) {
    const int t = n >> 4, idx = ((t * 64 + ((n & 15) >> 2) * 16 + e) << 2) + (n & 3);
    const float c0 = PB[idx], c1 = PB[idx + TO * 256], c2 = PB[idx + 2 * TO * 256], c3 = PB[idx + 3 * TO * 256];
    return ((c0 + c1) + c2) + c3;
}

// ROOT INFERENCE, register-resident (representation + policy head; the search discards the root's value, mcts.py:356-367): the same
// two ideas as the simulations' network -- wave w keeps its quarter of every num_planes-wide layer in accumulators, which ARE the B
// operands of the K-split layer behind it -- on the weights in their generic packed layout (mz_mlp.h), two blocks in flight.  The
// generic mlp_initial_tile writes both 512 x 16 hidden layers to LDS and reads them back (7 barriers); this form has 4, and the
// same summation order: blocks ascending, k-steps ascending, quarter chains ((c0 + c1) + c2) + c3.
//   `scr`: LDS scratch for the partial tiles, [4 waves][4 tiles] float4[64]; `hid0`: this lane's MFMA-side row of node 0 in the node
//   store (or null); every thread of the workgroup calls it (`active`: waves 0-3 of the 8-wave kernels), 4 barriers inside.
template <int NT>
__device__ __forceinline__ void root_inference_fast(const MlpNet& net, const MlpLds& o, float* lds, float4* scr, float* hid0, float* pi0, int tid, int wave,
                                                    int lane, bool active) {
    const int q = lane >> 4;
    const float* bias = lds;
    MZ_ROOT_TS_START();
    f32x4 h1[NT];
    if (active) {
        // ---- representation layer 1: X (observation, packed in LDS) -> this wave's NT tiles ----
        const MlpLayer& L0 = net.L[L_REP0];
        const float4* wp = reinterpret_cast<const float4*>(L0.w) + (size_t)(NT * wave) * L0.kg * 64 + lane;
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + L0.b_lds + (NT * wave + j) * 16 + q * 4);
            h1[j] = f32x4{bv.x, bv.y, bv.z, bv.w};
        }
        float4 w0[NT], w1[NT];
#pragma unroll
        for (int j = 0; j < NT; j++) w0[j] = ldg4(&wp[(size_t)j * L0.kg * 64]);
        for (int g = 0; g < L0.kg; g += 2) {  // two blocks per iteration: the second one's weights are in flight while the first multiplies
            const int g1 = g + 1 < L0.kg ? g + 1 : g, g2 = g + 2 < L0.kg ? g + 2 : g;
#pragma unroll
            for (int j = 0; j < NT; j++) w1[j] = ldg4(&wp[((size_t)j * L0.kg + g1) * 64]);
            mma_block<NT>(h1, w0, reinterpret_cast<const float4*>(lds + o.X)[g * 64 + lane], g + 1 < L0.kg ? 4 : L0.last_steps);
#pragma unroll
            for (int j = 0; j < NT; j++) w0[j] = ldg4(&wp[((size_t)j * L0.kg + g2) * 64]);
            if (g + 1 < L0.kg) mma_block<NT>(h1, w1, reinterpret_cast<const float4*>(lds + o.X)[(g + 1) * 64 + lane], g + 2 < L0.kg ? 4 : L0.last_steps);
        }
        relu_tiles<NT>(h1);
        MZ_ROOT_TS(0);
        // ---- representation layer 2, K-split: this wave's quarter (blocks NT w .. NT w + NT - 1) of all four output tiles ----
        const MlpLayer& L1 = net.L[L_REP1];
        const float4* wq = reinterpret_cast<const float4*>(L1.w) + lane;
        f32x4 acc2[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + L1.b_lds + t * 16 + q * 4);
            acc2[t] = wave == 0 ? f32x4{bv.x, bv.y, bv.z, bv.w} : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        float4 wa[4], wb[4];
#pragma unroll
        for (int t = 0; t < 4; t++) wa[t] = ldg4(&wq[((size_t)t * L1.kg + NT * wave) * 64]);
#pragma unroll
        for (int kk = 0; kk < NT; kk += 2) {
#pragma unroll
            for (int t = 0; t < 4; t++) wb[t] = ldg4(&wq[((size_t)t * L1.kg + NT * wave + (kk + 1 < NT ? kk + 1 : kk)) * 64]);
#pragma unroll
            for (int st = 0; st < 4; st++) {
#pragma unroll
                for (int t = 0; t < 4; t++) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(wa[t], st), h1[kk][st], acc2[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; t++) wa[t] = ldg4(&wq[((size_t)t * L1.kg + NT * wave + (kk + 2 < NT ? kk + 2 : kk)) * 64]);
            if (kk + 1 < NT) {
#pragma unroll
                for (int st = 0; st < 4; st++) {
#pragma unroll
                    for (int t = 0; t < 4; t++) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(wb[t], st), h1[kk + 1][st], acc2[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) scr[(wave * 4 + t) * 64 + lane] = make_float4(acc2[t][0], acc2[t][1], acc2[t][2], acc2[t][3]);
    }
    __syncthreads();
    MZ_ROOT_TS(1);
    if (active) {
        // ---- un-normalised state from the partials, min / max, this wave's tile normalised -> HS (LDS) and node 0 of the node store ----
        // (four named tiles, not an array: `hw = h[wave]` on an array becomes a load through a selected address, i.e. scratch memory)
        auto tile = [&](int t) {
            const float4 c0 = scr[(0 * 4 + t) * 64 + lane], c1 = scr[(1 * 4 + t) * 64 + lane], c2 = scr[(2 * 4 + t) * 64 + lane], c3 = scr[(3 * 4 + t) * 64 + lane];
            return f32x4{((c0.x + c1.x) + c2.x) + c3.x, ((c0.y + c1.y) + c2.y) + c3.y, ((c0.z + c1.z) + c2.z) + c3.z, ((c0.w + c1.w) + c2.w) + c3.w};
        };
        const f32x4 h0 = tile(0), h1_ = tile(1), h2 = tile(2), h3 = tile(3);
        float mn = h0[0], mx = h0[0];
#pragma unroll
        for (int r = 0; r < 4; r++) { mn = fmin2(h0[r], mn); mx = fmax2(h0[r], mx); }
#pragma unroll
        for (int r = 0; r < 4; r++) { mn = fmin2(h1_[r], mn); mx = fmax2(h1_[r], mx); }
#pragma unroll
        for (int r = 0; r < 4; r++) { mn = fmin2(h2[r], mn); mx = fmax2(h2[r], mx); }
#pragma unroll
        for (int r = 0; r < 4; r++) { mn = fmin2(h3[r], mn); mx = fmax2(h3[r], mx); }
        mn = rows_min(mn);
        mx = rows_max(mx);
        const float d = (mx - mn) + 1e-8f;
        f32x4 hw = h0;
        if (wave == 1) hw = h1_;
        if (wave == 2) hw = h2;
        if (wave == 3) hw = h3;
        const float4 hs = make_float4((hw[0] - mn) / d, (hw[1] - mn) / d, (hw[2] - mn) / d, (hw[3] - mn) / d);
        reinterpret_cast<float4*>(lds + o.HS)[wave * 64 + lane] = hs;
        if (hid0) *reinterpret_cast<float4*>(hid0 + wave * 16 + q * 4) = hs;
    }
    __syncthreads();
    MZ_ROOT_TS(2);
    if (active) {
        // ---- policy head layer 1 (normalised state from LDS) and layer 2 (K-split, one output tile: A <= 16) ----
        const MlpLayer& L2 = net.L[L_POL0];
        const float4* wp = reinterpret_cast<const float4*>(L2.w) + (size_t)(NT * wave) * L2.kg * 64 + lane;
        f32x4 p1[NT];
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + L2.b_lds + (NT * wave + j) * 16 + q * 4);
            p1[j] = f32x4{bv.x, bv.y, bv.z, bv.w};
        }
        float4 w0[NT], w1[NT];
#pragma unroll
        for (int j = 0; j < NT; j++) w0[j] = ldg4(&wp[(size_t)j * L2.kg * 64]);
#pragma unroll
        for (int g = 0; g < 4; g += 2) {  // hidden_dim 64: four full blocks
#pragma unroll
            for (int j = 0; j < NT; j++) w1[j] = ldg4(&wp[((size_t)j * L2.kg + g + 1) * 64]);
            mma_block_full<NT>(p1, w0, reinterpret_cast<const float4*>(lds + o.HS)[g * 64 + lane]);
#pragma unroll
            for (int j = 0; j < NT; j++) w0[j] = ldg4(&wp[((size_t)j * L2.kg + (g + 2 < 4 ? g + 2 : g)) * 64]);
            mma_block_full<NT>(p1, w1, reinterpret_cast<const float4*>(lds + o.HS)[(g + 1) * 64 + lane]);
        }
        relu_tiles<NT>(p1);
        MZ_ROOT_TS(3);
        const MlpLayer& L3 = net.L[L_POL1];
        const float4* wq = reinterpret_cast<const float4*>(L3.w) + (size_t)(NT * wave) * 64 + lane;  // tile 0, blocks NT w ..
        const float4 bv = *reinterpret_cast<const float4*>(bias + L3.b_lds + q * 4);
        f32x4 accp = wave == 0 ? f32x4{bv.x, bv.y, bv.z, bv.w} : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        float4 wv[NT];
#pragma unroll
        for (int kk = 0; kk < NT; kk++) wv[kk] = ldg4(&wq[kk * 64]);
#pragma unroll
        for (int kk = 0; kk < NT; kk++) {
#pragma unroll
            for (int st = 0; st < 4; st++) accp = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(wv[kk], st), p1[kk][st], accp, 0, 0, 0);
        }
        scr[wave * 64 + lane] = make_float4(accp[0], accp[1], accp[2], accp[3]);
    }
    __syncthreads();
    MZ_ROOT_TS(4);
    if (active && wave == 0) {  // the four quarter chains -> logits LG[e][n]
        const float4 c0 = scr[lane], c1 = scr[64 + lane], c2 = scr[128 + lane], c3 = scr[192 + lane];
        float* pl = lds + o.LG + (lane & 15) * o.lg_stride + q * 4;
        pl[0] = ((c0.x + c1.x) + c2.x) + c3.x; pl[1] = ((c0.y + c1.y) + c2.y) + c3.y;
        pl[2] = ((c0.z + c1.z) + c2.z) + c3.z; pl[3] = ((c0.w + c1.w) + c2.w) + c3.w;
    }
    __syncthreads();
    if (active) row_softmax(lds + o.LG + (tid >> 4) * o.lg_stride, pi0 + (tid >> 4) * net.A, net.A, tid & 15);
    __syncthreads();
    MZ_ROOT_TS(5);
}

// P = num_planes (256 or 512); TR / TV: tiles of the reward / value support (1 or 2)
// FUSE: device self-play with the environment inside this kernel (mz_selfplay_step on short moves)
// HW: HELPER WAVES.  The workgroup has 8 waves (two per SIMD; the kernel stays under 256 registers, so both fit): waves 0-3 are the
// kernel as ever, waves 4-7 take VALU work that does not have to wait for the MFMA stream of its own simulation -- the new state's
// normalisation + node-store write (while the reward head multiplies), the reward row's softmax (while the value head multiplies),
// the root's Dirichlet draws (while the root inference multiplies) -- and otherwise sleep at the workgroup barriers.  Same
// arithmetic, same results.  What this buys is LATENCY only: on gfx950 fp32 MFMAs and VALU instructions execute on the same
// ALUs (matrix fp32 peak == packed vector fp32 peak), so neither a lone wave (tools/micro/mfma_valu.hip: one independent v_fma_f32
// between two MFMAs costs +13 cycles, every further one +5; only LDS reads ride for free) nor a second wave of the SIMD (measured
// here: the reward head slows down by about the issue cycles of the normalisation that runs beside it) hides VALU ISSUE under
// MFMAs -- but the DPP / LDS / division latencies of those side jobs, which a lone wave sits through, overlap.  C2: -3.8 % with
// both jobs moved (731.0 -> 702.9 us); C3 (MSE heads: no softmax row) -1.2 % with the normalisation: the launcher picks
// (SearchParams::hwx).
// (Tried on top and measured, not kept: the hidden states of all nodes in the helper waves' REGISTERS -- v[64:239] indexed with
// s_set_gpr_idx from one inline-asm loop, 8 more nodes in the LDS the root buffers leave free, no HBM node store at all.  Correct,
// but the publish -> barrier -> indexed read -> LDS -> barrier chain costs what the MALL round trip of the HBM gather costs:
// 746.8 vs 744.7 us per C2 move.)
template <int P, int TR, int TV, bool FUSE = false, int AC = 0, bool HW = false, bool SPB = false>
__global__ __launch_bounds__(HW ? 2 * WG_THREADS : WG_THREADS) __attribute__((amdgpu_waves_per_eu(HW ? 2 : 1, HW ? 2 : 1))) void k_search_fast(const SearchParams Pm, const FastWeights FW) {
    // AC: what the launcher knows about the action count -- 2: exactly two actions, single player, categorical heads (classic control);
    // 10: exactly ten actions (TicTacToe: the backup's best-child refresh fully unrolled); 4: exactly four actions, categorical heads
    // (LunarLander's shape: the same unrolled refresh, the action count a constant in the tree phases); 0: anything up to 16
    constexpr bool TWO = AC == 2;
    constexpr int RD = fast_rd(P, AC);
    constexpr bool SC = AC == 10 && kFastSC, AX = AC == 10 && kFastAX;  // (the launcher sends only MSE-head nets to the SC build of AC == 10)
    using C = FastCfg<P, SC ? 0 : TR, SC ? 0 : TV, RD, AX ? 4 : 5>;
    constexpr int NT = C::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int wave8 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool main_w = !HW || wave8 < WG_WAVES;  // waves 0-3: the search as ever; waves 4-7 (HW): helpers
    const int tid = threadIdx.x & (WG_THREADS - 1), e = tid >> 4, a0 = tid & 15, lane = tid & 63;  // (helpers: their index among the helper threads)
    const int wave = wave8 & (WG_WAVES - 1);
    const int env_g = blockIdx.x * TILE_E + e;
    const bool env_ok = env_g < Pm.B;
    if constexpr (FUSE) { if (main_w) fused_env_pre(Pm, a0, env_g, env_ok); }
    const bool hwx = HW && (Pm.hwx & 1) != 0;   // helpers normalise the new state and write it to the node store
    const bool hwx2 = HW && (Pm.hwx & 2) != 0;  // helpers reduce the reward row
    const MlpNet& net = Pm.net;
    const int root_layers[4] = {L_REP0, L_REP1, L_POL0, L_POL1};
    const float warm = main_w ? prefetch_root_weights(net, root_layers, 4, tid) : 0.0f;
    const MlpLds& o = Pm.o;
    float* pi0 = reinterpret_cast<float*>(smem + Pm.t_pi0);
    const float** src = reinterpret_cast<const float**>(smem + Pm.t_ptr);
    float** dst = reinterpret_cast<float**>(smem + Pm.t_ptr) + 16;

    MZ_STAMP_DECL
    MZ_STAMP_START();
    // ---- tables, tree, root (identical to k_search) ----
    if (main_w) {
        stage_biases(net, lds, tid);
        if (!HW) tree2_init(smem, Pm, tid, env_ok, env_g);  // (this kernel is tree_mode 2 only: the launcher sends other layouts to k_search)
        if (a0 == 0) {
            src[e] = env_ok ? Pm.obs + (size_t)env_g * net.in_dim : nullptr;
            dst[e] = env_ok ? Pm.hidden + (size_t)env_g * Pm.NN * net.H : nullptr;
        }
    }
    __syncthreads();
    MZ_STAMP(2);  // root: bias staging + tree tables
    // HW: the helpers draw the Dirichlet noise (into the t_tmp rows, read by the root prior after the inference's barriers) and fill
    // the tree's tables while waves 0-3 load the observation and run the root inference
    if (!HW) root_noise_lanes(smem, Pm, e, a0, env_g, env_ok);
    if (main_w) load_obs(net, lds + o.X, src, tid);
    __syncthreads();
    MZ_STAMP(15);  // root: Dirichlet draws + observation load
    if (HW && !main_w) {  // HW: the tree's tables (50 KiB of LDS writes) and the noise are the helpers' job, under the root inference
        tree2_init(smem, Pm, tid, env_ok, env_g);
        root_noise_lanes(smem, Pm, e, a0, env_g, env_ok);
    }
    {
        // (root_inference_fast needs a policy head of one tile and full hidden blocks: every shape this kernel is launched for)
        const int e2r = lane & 15, env2r = blockIdx.x * TILE_E + e2r;
        float* hid0 = (main_w && env2r < Pm.B) ? Pm.hidden + (size_t)env2r * Pm.NN * 64 : nullptr;
        root_inference_fast<NT>(net, o, lds, reinterpret_cast<float4*>(lds + o.H1), hid0, pi0, tid, wave, lane, main_w);  // the root's value is discarded (mcts.py:356-367)
    }
    if (Pm.S < 0) Pm.hidden[0] = warm;  // never true: keeps the prefetch loads alive
    __syncthreads();
    {
        MZ_ROOT_TS_START();
        SearchParams Pr = Pm;  // (the action count as a constant where the build knows it, as in the simulation loop)
        if constexpr (AC == 10 || AC == 4) Pr.A = AC;
        if constexpr (TWO) Pr.A = 2;
        if constexpr (SPB) { Pr.noise_mode = 2; Pr.has_mask = 1; }  // (self-play settings as constants, see the simulation loop)
        if (main_w && env_ok) root_prior_group(smem, Pr, e, a0, env_g);  // (A <= 16 in this kernel)
        MZ_ROOT_TS(6);
    }

    // ---- this wave's weight stream; ring primed with the first RD - 1 slots ----
    const int voff = lane * 16;
    WSrc ws = make_wsrc(FW.stream, FW.bytes, wave * C::SL * NT * 1024);
    ws.cur = ws.base + (RD - 1) * NT * 1024;  // the ring is primed with slots 0 .. RD-2
    float4 ring[RD][NT];
    if (main_w) {
#pragma unroll
        for (int i = 0; i < RD - 1; i++) {
#pragma unroll
            for (int j = 0; j < NT; j++) ring[i][j] = bload(ws, voff, i * NT + j);
        }
    }
    const int x_last = AX ? 4 : (TWO ? 1 : net.L[L_DYN0].last_steps);  // k-steps of the action block (1..4); AX: the last block is a hidden block
    // MFMA-side env of this lane (D column) and its hidden-state rows in the HBM node store
    const int e2 = lane & 15, q = lane >> 4;
    const int env2 = blockIdx.x * TILE_E + e2;
    const bool env2_ok = env2 < Pm.B;
    // (32-bit byte offsets from the uniform base, not pointers: one VGPR each across the simulation loop instead of two -- the two-action
    // FUSE build spilled the MFMA-side pointer and reloaded it behind a vmcnt(0) every simulation: C2 696.5 -> 691.5 us per move, C3
    // 228.7 -> 229.5, same box; the launcher checks that the node store is smaller than 4 GiB)
    const unsigned hid_sel = (unsigned)((env_ok ? env_g : 0) * Pm.NN * 64 + a0 * 4) * 4u;                  // select-side env (tid >> 4)
    const unsigned hid_mma = (unsigned)((env2_ok ? env2 : 0) * Pm.NN * 64 + wave * 16 + q * 4) * 4u;       // MFMA-side env (lane & 15)
    char* const hid_base = reinterpret_cast<char*>(Pm.hidden);
    const float* bias = lds;  // biases live in LDS (stage_biases)
    // SC: this lane's 4 NT weights of each scalar head -- row 0 of the packed second layer, input neurons 16 (NT w + j) + 4 q + i --
    // stay in registers for the whole move
    float4 w_sr[NT], w_sv[NT];
    if constexpr (SC) {
        if (main_w) {
            const float4* wr = reinterpret_cast<const float4*>(net.L[L_REW1].w);
            const float4* wv = reinterpret_cast<const float4*>(net.L[L_VAL1].w);
#pragma unroll
            for (int j = 0; j < NT; j++) {
                w_sr[j] = ldg4(&wr[(NT * wave + j) * 64 + q * 16]);
                w_sv[j] = ldg4(&wv[(NT * wave + j) * 64 + q * 16]);
            }
        }
    }
    // a scalar head's second layer on this wave's quarter (SC): lane (e, q) runs ONE fmaf chain over its own 4 NT neurons (tiles
    // ascending, registers ascending; the chain of wave 0, q 0 starts from the bias), the four q meet as (p0 + p1) + (p2 + p3)
    // -- two row-swap butterflies, the same sum in every lane -- and the waves' sums meet in LDS like any K-split tile,
    // ((c0 + c1) + c2) + c3: the order of mz_mlp.h's scalar_head_tile (C3: -2.9 %)
    auto scalar_head = [&](const f32x4 (&x)[NT], const float4 (&w)[NT], float b0, float* part /* [4 waves][256] floats, head_logit<1> layout */) {
        float pacc = (wave == 0 && q == 0) ? b0 : 0.0f;
#pragma unroll
        for (int j = 0; j < NT; j++) {
            pacc = fmaf(x[j][0], w[j].x, pacc); pacc = fmaf(x[j][1], w[j].y, pacc);
            pacc = fmaf(x[j][2], w[j].z, pacc); pacc = fmaf(x[j][3], w[j].w, pacc);
        }
        u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(pacc), __float_as_uint(pacc), false, false);
        pacc = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        r = __builtin_amdgcn_permlane32_swap(__float_as_uint(pacc), __float_as_uint(pacc), false, false);
        pacc = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        if (q == 0) part[wave * 256 + e2 * 4] = pacc;
    };
    // K-split partial tiles (LDS, aliased onto the H1 / V1 buffers that only the root inference uses):
    // PB [4 waves][4 tiles] dynamics layer 2, PBr [4][TR] reward layer 2, PBv [4][TV] value layer 2, float4[64] each
    float4* const PB = reinterpret_cast<float4*>(lds + o.H1);
    float4* const PBr = reinterpret_cast<float4*>(lds + o.V1);
    float4* const PBv = PBr + 4 * TR * 64;
    float* const rew_x = lds + o.OUT;  // HW: the helpers' reward scalars, [16 envs] (the OUT block is free: the root's value head is not evaluated)
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 xs0[5] = {zero4, zero4, zero4, zero4, zero4};
    NoHook nohook;
    // quarter chains 1..3 of a K-split layer start from +0, quarter 0 from the bias: one pointer select per layer instead
    // of a branch per component (the PM block is free in this kernel: 128 zeros)
    float* const zeros = lds + o.PM;
    if (main_w && tid < 128) zeros[tid] = 0.0f;
    const float* const b_d2 = wave == 0 ? bias + net.L[L_DYN1].b_lds : zeros;
    const float* const b_r2 = wave == 0 ? bias + net.L[L_REW1].b_lds : zeros;
    const float* const b_v2 = wave == 0 ? bias + net.L[L_VAL1].b_lds : zeros;
    __syncthreads();
    MZ_STAMP(0);  // root: tables + initial inference + prior

    int resume = 0;
    const int cp0 = env_ok ? Pm.cur[env_g] : 0, op0 = env_ok ? Pm.opp[env_g] : 0;  // the root's players: read once per move, not per descent
    // The tree phases' copy of the parameters, with what the build knows made a constant: the action count has 17 uses in mz_tree2.h
    // (row strides of the entry table, loop bounds, lane predicates), each a scalar load + multiply from the kernel argument otherwise
    // (C3 -1.2 %; pinning the LDS table offsets and the node count in registers on top of it -- hipcc re-loads ~30 kernel arguments at
    // the top of every simulation's select -- measured the same)
    SearchParams Pl = Pm;
    if constexpr (AC == 10 || AC == 4) Pl.A = AC;
    if constexpr (TWO) Pl.A = 2;
    // SPB: the board games' self-play settings as constants too -- two players, known bounds, discount 1 (its products fold away),
    // Dirichlet noise, tie and action draws from the device streams, sampled play over a legal-move mask: the runtime forms of these cost
    // the ten-action build 2.4 % (the launcher
    // checks every one of them and falls back to the build without the constants)
    if constexpr (SPB) { Pl.board = 1; Pl.has_bounds = 1; Pl.noise_mode = 2; Pl.rng_mode = 1; Pl.discount = 1.0; }
    // (the two-action build's counterpart -- classic control's self-play: no known bounds, device streams, sampled play -- measured
    // +0.5 % on C2, i.e. worse: that build is register-bound and the constants only moved its schedule)
    Tree2Env T;  // tree_mode 2: the env's search state lives in its lanes' registers
    tree2_env_init(T, Pl);
    // normalisation (util.py:31-36) of this wave's tile of h -> LDS (value head input) and the HBM node store
    // (the four tiles by value: through a reference to the array the `hw = h[wave]` selects below become ONE load through a selected
    // address, which pins the array in scratch memory -- a round trip per simulation, measured in the ISA)
    auto normalise_store = [&](const f32x4 h0, const f32x4 h1_, const f32x4 h2, const f32x4 h3, int s) {
        // min / max over the 64 features: 16 in this lane, the rest in the lanes of the other three rows (same env)
        float mn = h0[0], mx = h0[0];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            mn = fmin2(h0[r], mn); mx = fmax2(h0[r], mx);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            mn = fmin2(h1_[r], mn); mx = fmax2(h1_[r], mx);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            mn = fmin2(h2[r], mn); mx = fmax2(h2[r], mx);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            mn = fmin2(h3[r], mn); mx = fmax2(h3[r], mx);
        }
        mn = rows_min(mn);
        mx = rows_max(mx);
        const float d = (mx - mn) + 1e-8f;
        f32x4 hw = h0;
        if (wave == 1) hw = h1_;
        if (wave == 2) hw = h2;
        if (wave == 3) hw = h3;
        const float4 hs = make_float4((hw[0] - mn) / d, (hw[1] - mn) / d, (hw[2] - mn) / d, (hw[3] - mn) / d);
        reinterpret_cast<float4*>(lds + o.HS)[wave * 64 + lane] = hs;
        if (env2_ok) *reinterpret_cast<float4*>(hid_base + (size_t)(hid_mma + (unsigned)(s + 1) * 256u)) = hs;
    };
    constexpr bool MSE = AC == 10;  // the launcher sends only nets with MSE heads (reward / value support 1: TicTacToe's) to the ten-action build
    // un-normalised state h (64 x 16) from the four waves' partial tiles, ((c0 + c1) + c2) + c3, in two stages: wave w sums tile w
    // and publishes it (HN block), one more barrier, every wave reads the four finished tiles -- 8 tile reads per wave (4 per
    // helper wave) instead of 16: the one-stage form moved 128 KiB through the LDS port right behind the barrier (C2 -0.6 %, C3 -0.5 %)
    float4* const HT = reinterpret_cast<float4*>(lds + o.HN);
    auto reduce_tile = [&]() {
        const float4 c0 = PB[(0 * 4 + wave) * 64 + lane], c1 = PB[(1 * 4 + wave) * 64 + lane], c2 = PB[(2 * 4 + wave) * 64 + lane],
                     c3 = PB[(3 * 4 + wave) * 64 + lane];
        HT[wave * 64 + lane] = make_float4(((c0.x + c1.x) + c2.x) + c3.x, ((c0.y + c1.y) + c2.y) + c3.y, ((c0.z + c1.z) + c2.z) + c3.z,
                                           ((c0.w + c1.w) + c2.w) + c3.w);
    };
    auto read_h = [&](f32x4 (&h)[4]) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float4 v = HT[t * 64 + lane];
            h[t] = f32x4{v.x, v.y, v.z, v.w};
        }
    };
    const bool cat_heads = !MSE && (TWO || AC == 4 || (net.Sr != 1 && net.Sv != 1));  // both heads categorical (TWO, AC == 4: guaranteed by the launcher)
    if (HW && !main_w) {
        // ================= helper waves: the same barriers per simulation as waves 0-3 =================
        for (int s = 0; s < Pm.S; s++) {
            __syncthreads();  // B0 (X ready)
            __syncthreads();  // B1 (dynamics partial tiles ready)
            __syncthreads();  // B1b (summed tiles ready)
            if (hwx) {
                f32x4 h[4];
                read_h(h);
                normalise_store(h[0], h[1], h[2], h[3], s);
            }
            __syncthreads();  // B2 (reward partial tiles ready; HS written)
            if (hwx2 && cat_heads) {  // the reward row: softmax -> expectation -> signed_parabolic (util.py:70-93), 16 lanes per env
                const float* fr = reinterpret_cast<const float*>(PBr);
                const bool r0 = a0 < net.Sr, r1 = a0 + 16 < net.Sr;
                const float lr0 = head_logit<TR>(fr, r0 ? a0 : 0, e), lr1 = head_logit<TR>(fr, r1 ? a0 + 16 : 0, e);
                const float rew = row2_logits_to_scalar(lr0, lr1, r0, r1, net.Sr, a0);
                if (a0 == 0) rew_x[e] = rew;
            }
            __syncthreads();  // B3 (value partial tiles ready; reward scalars written)
        }
        return;
    }
    for (int s = 0; s < Pm.S; s++) {
        tree2_select<false, TWO ? 2 : 0>(smem, Pl, tid, env_ok, env_g, T, cp0, op0, resume);
        const int lp = T.lp, la = T.la;
        // gather: the env's 16 lanes fetch the parent's hidden state (64 floats = one float4 per lane) straight after
        // their descent and store it, with the one-hot action (network.py:191-193), into the packed B-operand buffer
        {
            float* X = lds + o.X;
            float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (env_ok) hv = *reinterpret_cast<const float4*>(hid_base + (size_t)(hid_sel + (unsigned)lp * 256u));
            reinterpret_cast<float4*>(X)[(a0 >> 2) * 64 + (a0 & 3) * 16 + e] = hv;  // pk(4 a0 .. 4 a0 + 3, e)
            if constexpr (AX) { if (a0 == 0) reinterpret_cast<int*>(smem + Pm.t_sel)[e * 4 + 1] = env_ok ? la : 0; }
            else X[pk_act(a0, e, 4)] = (a0 == la && a0 < Pm.A) ? 1.0f : 0.0f;
        }
        __syncthreads();
        MZ_STAMP(1);  // select + gather
        // ---- dynamics layer 1 (this wave's NT tiles; X from LDS) and layer 2 (K-split partial from registers) ----
        f32x4 h1[NT], h[4];
        {
            const float* b1 = bias + net.L[L_DYN0].b_lds + wave * NT * 16;
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const float4 bv = *reinterpret_cast<const float4*>(b1 + j * 16 + q * 4);
                h1[j] = f32x4{bv.x, bv.y, bv.z, bv.w};
            }
            float4 xs[5];
#pragma unroll
            for (int g = 0; g < C::XG; g++) xs[g] = reinterpret_cast<const float4*>(lds + o.X)[g * 64 + lane];
            float4 wa4[NT];
            if constexpr (AX) {  // the action's weight column for this lane's 4 NT neurons (requested here, added behind the MFMAs)
                const int la2 = reinterpret_cast<const int*>(smem + Pm.t_sel)[e2 * 4 + 1];
                const float4* wc = reinterpret_cast<const float4*>(FW.wact + la2 * P + wave * NT * 16 + q * 4);
#pragma unroll
                for (int j = 0; j < NT; j++) wa4[j] = wc[j * 4];
            }
            WideSlots<C, RD, C::I_D1, 0, C::XG, true>::run(ring, ws, voff, xs, h, x_last, h1, nohook);
            if constexpr (AX) {  // x = one-hot: the chain's last A steps are fmaf(0, w, acc) = acc and one fmaf(1, w, acc) = acc + w
#pragma unroll
                for (int j = 0; j < NT; j++) { h1[j][0] += wa4[j].x; h1[j][1] += wa4[j].y; h1[j][2] += wa4[j].z; h1[j][3] += wa4[j].w; }
            }
            relu_tiles<NT>(h1);
            MZ_STAMP(3);  // dynamics layer 1
            f32x4 acc2[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float4 bv = *reinterpret_cast<const float4*>(b_d2 + t * 16 + q * 4);
                acc2[t] = f32x4{bv.x, bv.y, bv.z, bv.w};
            }
            SplitSlots<C, RD, C::I_D2, 0, 4>::run(ring, ws, voff, h1, acc2, nohook);
#pragma unroll
            for (int t = 0; t < 4; t++) PB[(wave * 4 + t) * 64 + lane] = make_float4(acc2[t][0], acc2[t][1], acc2[t][2], acc2[t][3]);
        }
        __syncthreads();
        MZ_STAMP(4);  // dynamics layer 2 (partials)
        // ---- every wave needs the whole un-normalised state h (64 x 16): two-stage exchange (above); normalisation
        // (util.py:31-36) of its own tile goes to LDS (value head input) and to the HBM node store (HW: by the helpers) ----
        reduce_tile();
        __syncthreads();
        read_h(h);
        if (!hwx) normalise_store(h[0], h[1], h[2], h[3], s);
        MZ_STAMP(5);  // reduce + normalise + hidden store
        // ---- reward head on the UN-normalised state (network.py:195-196): layer 1 from registers, layer 2 K-split ----
        {
            f32x4 r1[NT];
            const float* b1 = bias + net.L[L_REW0].b_lds + wave * NT * 16;
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const float4 bv = *reinterpret_cast<const float4*>(b1 + j * 16 + q * 4);
                r1[j] = f32x4{bv.x, bv.y, bv.z, bv.w};
            }
            WideSlots<C, RD, C::I_R1, 0, 4, false>::run(ring, ws, voff, xs0, h, 4, r1, nohook);
            relu_tiles<NT>(r1);
            if constexpr (SC) {
                scalar_head(r1, w_sr, bias[net.L[L_REW1].b_lds], reinterpret_cast<float*>(PBr));
            } else {
            f32x4 accr[TR];
#pragma unroll
            for (int t = 0; t < TR; t++) {
                const float4 bv = *reinterpret_cast<const float4*>(b_r2 + t * 16 + q * 4);
                accr[t] = f32x4{bv.x, bv.y, bv.z, bv.w};
            }
            SplitSlots<C, RD, C::I_R2, 0, TR>::run(ring, ws, voff, r1, accr, nohook);
#pragma unroll
            for (int t = 0; t < TR; t++) PBr[(wave * TR + t) * 64 + lane] = make_float4(accr[t][0], accr[t][1], accr[t][2], accr[t][3]);
            }
        }
        __syncthreads();
        MZ_STAMP(6);  // reward head
        // ---- value head on the normalised state (from LDS) ----
        {
            f32x4 v1[NT];
            const float* b1 = bias + net.L[L_VAL0].b_lds + wave * NT * 16;
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const float4 bv = *reinterpret_cast<const float4*>(b1 + j * 16 + q * 4);
                v1[j] = f32x4{bv.x, bv.y, bv.z, bv.w};
            }
            float4 xs[5];
#pragma unroll
            for (int g = 0; g < 4; g++) xs[g] = reinterpret_cast<const float4*>(lds + o.HS)[g * 64 + lane];
            xs[4] = zero4;
            WideSlots<C, RD, C::I_V1, 0, 4, true>::run(ring, ws, voff, xs, h, 4, v1, nohook);
            relu_tiles<NT>(v1);
            if constexpr (SC) {
                scalar_head(v1, w_sv, bias[net.L[L_VAL1].b_lds], reinterpret_cast<float*>(PBv));
            } else {
            f32x4 accv[TV];
#pragma unroll
            for (int t = 0; t < TV; t++) {
                const float4 bv = *reinterpret_cast<const float4*>(b_v2 + t * 16 + q * 4);
                accv[t] = f32x4{bv.x, bv.y, bv.z, bv.w};
            }
            SplitSlots<C, RD, C::I_V2, 0, TV>::run(ring, ws, voff, v1, accv, nohook);
#pragma unroll
            for (int t = 0; t < TV; t++) PBv[(wave * TV + t) * 64 + lane] = make_float4(accv[t][0], accv[t][1], accv[t][2], accv[t][3]);
            }
            PadSlots<C, RD, C::I_END>::run(ring, ws, voff);
        }
        // the backup's tree reads that do not depend on this simulation's reward / value: issued ahead of the barrier and the softmax
        // (two-action searches: -0.9 % on C2; with ten actions the backup is dominated by its refresh loop and the move was +0.6 %)
        Backup2Pre bpre;
        if constexpr (TWO) bpre = tree2_backup_prefetch<2>(smem, Pl, tid, env_ok, s, T);
        __syncthreads();
        MZ_STAMP(7);  // value head
        // softmax -> expectation -> signed_parabolic (util.py:70-93) in registers: 16 lanes per row, 2 logits per lane;
        // the results are segment-uniform, so the env's lanes go straight on to expand + backup
        {
            // (a lone wave cannot hide this under its own MFMAs -- tools/micro/mfma_valu.hip; HW: the reward row is the helpers')
            const float* fr = reinterpret_cast<const float*>(PBr);
            const float* fv = reinterpret_cast<const float*>(PBv);
            float rew, val;
            if constexpr (MSE) {
                rew = head_logit<TR>(fr, 0, e);
                val = head_logit<TV>(fv, 0, e);
            } else
            if (hwx2 && cat_heads) {
                const bool v0 = a0 < net.Sv, v1 = a0 + 16 < net.Sv;
                const float lv0 = head_logit<TV>(fv, v0 ? a0 : 0, e), lv1 = head_logit<TV>(fv, v1 ? a0 + 16 : 0, e);
                rew = rew_x[e];
                val = row2_logits_to_scalar(lv0, lv1, v0, v1, net.Sv, a0);
            } else if (cat_heads) {  // (every single-player configuration): the two rows interleaved
                const bool hr0 = a0 < net.Sr, hr1 = a0 + 16 < net.Sr, hv0 = a0 < net.Sv, hv1 = a0 + 16 < net.Sv;
                const float l0[2] = {head_logit<TR>(fr, hr0 ? a0 : 0, e), head_logit<TV>(fv, hv0 ? a0 : 0, e)};
                const float l1[2] = {head_logit<TR>(fr, hr1 ? a0 + 16 : 0, e), head_logit<TV>(fv, hv1 ? a0 + 16 : 0, e)};
                const bool h0[2] = {hr0, hv0}, h1[2] = {hr1, hv1};
                const int SS[2] = {net.Sr, net.Sv};
                float out[2];
                rows2_logits_to_scalars(l0, l1, h0, h1, SS, a0, out);
                rew = out[0];
                val = out[1];
            } else {
            if (net.Sr == 1) {
                rew = head_logit<TR>(fr, 0, e);
            } else {
                const bool r0 = a0 < net.Sr, r1 = a0 + 16 < net.Sr;
                const float lr0 = head_logit<TR>(fr, r0 ? a0 : 0, e), lr1 = head_logit<TR>(fr, r1 ? a0 + 16 : 0, e);
                rew = row2_logits_to_scalar(lr0, lr1, r0, r1, net.Sr, a0);
            }
            if (net.Sv == 1) {
                val = head_logit<TV>(fv, 0, e);
            } else {
                const bool v0 = a0 < net.Sv, v1 = a0 + 16 < net.Sv;
                const float lv0 = head_logit<TV>(fv, v0 ? a0 : 0, e), lv1 = head_logit<TV>(fv, v1 ? a0 + 16 : 0, e);
                val = row2_logits_to_scalar(lv0, lv1, v0, v1, net.Sv, a0);
            }
            }
            MZ_STAMP(8);  // softmax + expectation + transform
            resume = tree2_backup<TWO ? 2 : 0, (AC > 2 ? AC : 0), TWO>(smem, Pl, tid, env_ok, s, rew, val, T, &bpre);  // backup and the next select of an env run on the same 16 lanes: no barrier
        }
        MZ_STAMP(9);  // expand + backup
    }
    // The tail reads its parameters (output pointers, mask, temperatures; FUSE: the env's two dozen pointers) through a pointer to the
    // kernel-argument segment that is made opaque HERE: left to itself hipcc merges these loads with the ones at the top of the kernel
    // and carries ~100 scalars across the simulation loop -- 200+ SGPR spills in the FUSE instantiations, every reload a v_readlane
    // plus wait states on a wave that runs alone (1.6-2 k v_readlane in the kernel against 40-70 without the env, counted in the ISA).
    typedef const __attribute__((address_space(4))) SearchParams* late_params_t;
    late_params_t late = (late_params_t)__builtin_amdgcn_kernarg_segment_ptr();  // (SearchParams is the kernel's first argument)
    asm volatile("" : "+s"(late));
    const SearchParams& Pt = *(const SearchParams*)late;
    bool stepped = false;
    if constexpr (FUSE && TWO) {
        if (Pt.fenv.env.kind == ENV_CARTPOLE) {  // the step's inputs are requested before the play policy, see mz_env.h
            CartPolePre pre;
            const bool stepper = env_ok && a0 == 0;
            if (stepper) cartpole_prefetch(Pt.fenv, env_g, pre);
            int action = 0;
            double rootv = 0.0;
            if (env_ok) {
                SearchParams Pf = Pt;
                Pf.A = 2;
                tree2_finish_group(smem, Pf, e, a0, env_g, action, rootv);
            }
            if (stepper) cartpole_step_prefetched(Pt.fenv, env_g, pre, action, rootv, reinterpret_cast<const double*>(smem + Pt.t_tmp) + e * 2);
            stepped = true;
        }
    }
    if constexpr (FUSE && AC == 10) {
        if (Pt.fenv.env.kind == ENV_TICTACTOE && Pt.fenv.env.nn <= 16 && Pt.A <= 16) {  // small board: the same idea, see mz_env.h
            BoardSmallPre pre;
            if (env_ok) board_small_prefetch(Pt.fenv, env_g, a0, pre);
            int action = 0;
            double rootv = 0.0;
            if (env_ok) {
                SearchParams Pf = Pt;
                Pf.A = 10;
                if constexpr (SPB) { Pf.deterministic = 0; Pf.has_mask = 1; Pf.rng_mode = 1; }
                tree2_finish_group(smem, Pf, e, a0, env_g, action, rootv);
                board_step_small_prefetched(Pt.fenv, env_g, a0, pre, action, rootv, reinterpret_cast<const double*>(smem + Pt.t_tmp) + e * 10);
            }
            stepped = true;
        }
    }
    if (!stepped) {
        if (env_ok) tree2_finish_group(smem, Pt, e, a0, env_g);  // (A <= 16 in this kernel)
        if constexpr (FUSE)
            if (env_ok) env_step_group(Pt.fenv, env_g, a0);  // env.step + record + auto-reset with the action lane 0 just sampled
    }
    MZ_STAMP(10);  // play policy + action
    MZ_STAMP_FLUSH(Pm);
}

}  // namespace mz
