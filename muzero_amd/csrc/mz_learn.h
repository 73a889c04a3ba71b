// mz_learn.h -- the learner step of the MLP nets as hand-written gfx950 kernels.
//
// Reference: `calc_loss` pipeline.py:541-612, `loss_func` :615-629, the update loop :238-255 (Adam + MultiStepLR + optional
// clip_grad_norm_), target projection util.py:48-59,96-116, networks network.py:140-267.  No ATen on this path.
//
// DECOMPOSITION.  The batch is cut into tiles of 16 samples (the N dimension of v_mfma_f32_16x16x4_f32).  One update is a short
// sequence of launches on one stream; a dependent kernel boundary costs ~1.5 us on this part, less than any in-launch grid barrier
// (MI355X_MICROARCH.md, rows "boundary" / "barrier-xcd"), and every stage's workgroups depend only on workgroups of the SAME tile in
// the previous stage:
//   k_learn_repr            grid (tiles)        representation net forward -> h_0
//   k_learn_unroll          K launches, grid (tiles): role 0, dynamics_k forward (h_k, a_k -> u_{k+1}, h_{k+1}) -- the only serial chain of
//                                               the forward sweep;
//                           then ONE launch, grid (tiles, 3, K): roles 1 / 2: policy_k / value_k forward + loss + backward to dL/dh_k;
//                                               role 3: reward_k forward + loss + backward to dL/du_{k+1} -- every head of every step
//                                               needs only h_k / u_{k+1}, so all 3 K of them run side by side (120 workgroups at batch 128)
//   k_learn_back<k>         grid (tiles)        dL/dh_{k+1} (halved, pipeline.py:584) -> normalisation backward (+ reward's dL/du_{k+1})
//                                               -> dynamics_k backward -> dL/dh_k                                    (k = K-1 .. 0, then repr)
//   k_learn_dw              grid (jobs + 1, split)  every weight / bias gradient: dW = sum over (step, sample) dZ x^T as MFMA tiles; the extra
//                                               workgroup adds up the loss
// By batch size (the launcher, learner.hip mzl_grad): up to ~64 tiles both chains are cut across the planes (k_learn_fwd_sliced /
// k_learn_back_sliced: more workgroups than tiles); from 96 tiles on the K dynamics stages of a tile run inside two PERSISTENT kernels
// (k_learn_dyn_chain / k_learn_dyn_back_chain: operands loaded once per workgroup); the heads take a three-per-CU streaming build from
// 16 tiles on; long reductions use k_learn_dw_big + k_learn_gradsum.
//   k_learn_adam            grid (blocks, 20)   clip + Adam (L2 weight decay in the gradient) + re-pack of the MFMA operand copies
// Every head's loss gradient is known as soon as its logits are (dL/dz = (softmax - target) w / (B K)), so the heads run forward AND
// backward inside the forward sweep; the backward sweep is the dynamics chain alone.
//
// MFMA ORIENTATION.  All layer GEMMs are computed "transposed": A operand = activations (16 samples x 4 features from LDS),
// B operand = weights (4 features x 16 outputs), D[sample][output].  A lane (f = lane & 15, sq = lane >> 4) then holds output feature
// 16 t + f for samples 4 sq .. 4 sq + 3 -- which is exactly the A / B operand layout of the weight-gradient GEMM, whose reduction runs
// over SAMPLES: every tensor dW needs (layer inputs, pre-activation gradients) leaves its producer as one 16-byte store per lane
// ("T blocks": 1 KiB per (16 features x 16 samples)), and k_learn_dw streams them with one 16-byte load per lane and operand.
// The copy the NEXT layer multiplies lives in LDS in the planner's packed order pk(k, e) (mz_mlp.h), written with four ds_write_b32.
//
// Summation orders are this file's own (fp32 throughout); parity with the reference is by tolerance (tests/test_gpu_hip_learner.py:
// loss 1e-4, gradients 2e-3 relative, three optimizer steps), not bit-exactness -- the reference's own CPU GEMM order is unpinned.
#pragma once
#include <type_traits>

#include "mz_device.h"

namespace mzl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LW = 8;        // waves per workgroup
constexpr int LT = LW * 64;  // threads per workgroup
constexpr int TILE = 16;     // samples per tile
constexpr int RPS = LT / TILE;  // threads per sample in the row phases (32)
constexpr int DX_PARTS = 4;     // partial dL/dh tiles a backward stage may leave for the next one (k_learn_back_sliced)

__device__ __forceinline__ int pk(int k, int e) { return (((k >> 4) * 64 + ((k >> 2) & 3) * 16 + e) << 2) + (k & 3); }

enum { REP0 = 0, REP1, DYN0, DYN1, REW0, REW1, POL0, POL1, VAL0, VAL1, NLAYER };

struct LLayer {
    const float* wp;   // forward operand copy   [nt][kg][64][4]: W[16 t + (lane & 15)][col(16 g + 4 (lane >> 4) + i)]
    const float* wtp;  // transposed copy        [kg][nt][64][4]: W[16 g + 4 (lane >> 4) + i][col(16 t + (lane & 15))], t = INPUT tile, g = output block
    const float* b;    // [nt * 16], zero padded
    int n, k;          // real outputs / inputs
    int nt, kg;        // 16-tiles of the (padded) outputs / inputs
};

struct LNet {
    LLayer L[NLAYER];
    int in_dim, A, P, H, Sv, Sr, K;
    int in_t, h_t, a_t, p_t, sv_t, sr_t;  // 16-tiles of: observation, hidden state, actions, planes, value / reward support
    int lgs;                               // row stride of the logits rows in LDS (odd)
};

// tensors saved for the weight gradients, as T blocks: block (step, tile, feature tile t) at ((step * tiles + tile) * ft + t) * 256
struct LSave {
    float *in_rep, *h1_rep, *dz_rep0, *dz_rep1;  // observation, relu(layer 1), dL/dz1, dL/du_0
    float *x, *h1_dyn, *dz_dyn0, *dz_dyn1;       // [h_k | onehot(a_k)], relu, dL/dz1, dL/du_{k+1}
    float *u_in, *h1_rew, *dz_rew0, *dz_rew1;    // u_{k+1}, relu, dL/dz1, dL/dlogits
    float *h1_pol, *dz_pol0, *dz_pol1;
    float *h1_val, *dz_val0, *dz_val1;
    // chain tensors handed from stage to stage in pk order: [K + 1][tiles][h_t * 256]
    float *hc, *uc;                  // h_k (normalised), u_k (before normalisation)
    float *dxd, *dxp, *dxv, *dxr;    // dL/dh_k from dynamics_k / policy_k / value_k ; dL/du_{k+1} from reward_k   [K][tiles][h_t * 256]
                                     // (dxd: [K][tiles][DX_PARTS][h_t * 256] -- the sliced backward stage leaves one partial per slice)
    int dx_parts;                    // partials the backward stages of this step write and read (1: unsliced)
    float* up;                       // [K + 1][tiles][DX_PARTS][h_t * 256]: partial u_k (no bias) left by the sliced forward stages
    float* lossp;                    // [3 K][tiles] partial losses (sum over the tile of w * loss)
    long long* stamps;               // diagnostic (tools/dev/learn_stamps.py): cycle stamps of tile 0's workgroups, or nullptr
    int* actc;                       // [K][tiles][16] the batch's actions, gathered once by k_learn_repr (-1: no such sample)
};

struct LBatch {
    const void* state;     // replay ring [cap][in_dim] float32 or int8
    const void* action;    // [cap][K] int8 or int16
    const float* pi;       // [cap][K][A]
    const float* value;    // [cap][K]
    const float* reward;   // [cap][K]
    const int64_t* idx;    // [B] rows of the ring
    const float* w;        // [B] importance weights
    float* prio;           // [B] out: |v_0 - z_0| (pipeline.py:609)
    int B, tiles;
    int state_i8, action_bytes;
};

// ------------------------------------------------------------------------------------------------------------------
// GEMM pieces (transposed orientation, see header)
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ldg4(const float4* p) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    return make_float4(v[0], v[1], v[2], v[3]);
}

// the same load through a pointer that was itself LOADED from memory (the job tables of the weight-gradient kernels): the compiler
// cannot tell its address space and emits flat_load, which ticks lgkmcnt as well and drains out of order -- every wait becomes
// vmcnt(0) lgkmcnt(0).  An explicit global pointer keeps them counted vector-memory loads (k_learn_dw_big: -0.7 % at batch 16384).
__device__ __forceinline__ float4 ldg4g(const float4* p) {
    typedef const __attribute__((address_space(1))) f32x4* gptr_t;
    const f32x4 v = *(gptr_t)(p);
    return make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ f32x4 mfma4(const float4 x, const float4 w, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, w.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, w.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, w.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, w.w, acc, 0, 0, 0);
    return acc;
}

// The weight operands of a GEMM are loaded by one call and multiplied by another, so that the loads can be issued a whole phase
// early -- before the barrier, the LDS copies and the MFMAs of the PREVIOUS GEMM: an L2 round trip is ~1 us here, a layer's MFMAs 1-2 us,
// and a stage kernel is a chain of four such layers on one workgroup (un-prefetched, the stage spent half its time waiting for weights).
//
// Many output tiles, short reduction (first layers, and the backward pass through a second layer): output tiles are dealt to the
// waves round-robin, four per wave.  epi(t, acc): acc[i] = Y[16 t + (lane & 15)][sample 4 (lane >> 4) + i].
#ifndef MZL_GW
#define MZL_GW 2
#endif
constexpr int WKG = 6;  // reduction blocks the register-resident form holds (dynamics layer 1 of the benchmark nets: 5)
struct WideW {
    float4 w[WKG][4];
    float bv[4];  // this lane's bias value of each of its four tiles (requested with the weights: a load at the GEMM's head would
                  // have to wait for every load issued before it -- vmcnt retires in order -- i.e. for the prefetch itself)
    bool fast;  // nt <= 4 LW tiles and kg <= WKG blocks: everything is in registers; otherwise wide_mma loads as it goes
};
// (no run-time guards around the loads: `if (d < kg) load` is compiled to a branch around the load AND a wait for it -- one exposed
// L2 round trip per block; and a switch over block counts makes the compiler wait at the end of every case, because the cases'
// registers meet in phi copies.  So: a compile-time bound KGM per call site, all KGM blocks loaded, addresses clamped to the last
// real block -- a redundant load hits the line its neighbour has just fetched)
template <int KGM, bool BIAS, bool F>
__device__ __forceinline__ void wide_load(WideW& W, const float* __restrict__ wp, const float* __restrict__ bias, int nt, int kg, int wave, int lane) {
    static_assert(KGM <= WKG, "register-resident form holds WKG blocks");
    // F: the launcher has checked every GEMM of the net: register-resident operands, no generic code in the kernel; !F: the streaming
    // forms (two workgroups per CU under 128 VGPRs), except first layers of at most MZL_GW blocks, which fit that budget too
    W.fast = F || (KGM <= MZL_GW && nt <= 4 * LW && kg <= KGM);
    if (!W.fast) return;
    const float4* W4 = reinterpret_cast<const float4*>(wp) + lane;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int t = wave + j * LW, tc = t < nt ? t : nt - 1;
        W.bv[j] = 0.0f;
        if constexpr (BIAS) W.bv[j] = bias[tc * 16 + (lane & 15)];
#pragma unroll
        for (int d = 0; d < KGM; d++) W.w[d][j] = ldg4(W4 + ((size_t)tc * kg + (d < kg ? d : kg - 1)) * 64);
    }
}
// MEASURED AND NOT KEPT (round 4): the next GEMM's operand loads issued BETWEEN this GEMM's MFMAs (one or two per group of four,
// pinned with sched_barrier) instead of as a burst in front of them.  The burst holds a wave at the vector-memory issue for ~2 k
// cycles (8 waves x 16 KiB through the CU's 64 B / clk L1 path), but spreading the loads is WORSE (first layer 8.5 k -> 10.8 k cycles,
// the second layer's transpose 3.5 k -> 6.5 k): a wave issues in order, so a load waiting for the L1 path holds the MFMAs behind it.
// The stage kernels are bound by weight bytes per MFMA (16 samples per weight load), not by where the requests sit.
struct NoPf {
    __device__ __forceinline__ void operator()(int) const {}
};
constexpr int KSB = 16;  // reduction blocks per wave the register-resident form holds (512 planes, 4 output tiles: 16)
struct KsW {
    float4 w[KSB];
    float bv;  // bias of the tile this wave finishes (waves 0 .. nt-1), requested with the weights
    bool fast;
};
template <bool F, int KGM, bool LEAN = false, typename Epi, typename Pf>
__device__ __forceinline__ void wide_mma(const WideW& W, const float* __restrict__ wp, const float* __restrict__ bias, int nt, int kg, const float* Xs,
                                         int wave, int lane, Epi epi, Pf) {
    const float4* X4 = reinterpret_cast<const float4*>(Xs) + lane;
    if (F || W.fast) {
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = f32x4{W.bv[j], W.bv[j], W.bv[j], W.bv[j]};
#pragma unroll
        for (int d = 0; d < KGM; d++)
            if (d < kg) {  // (ONE guarded block per reduction block: with a guard per accumulator the four MFMA chains end up in four
                           // basic blocks and cannot be interleaved -- first layer 8.3 k -> 10.6 k cycles, measured)
                const float4 x = X4[d * 64];
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = mfma4(x, W.w[d][j], acc[j]);
            }
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (wave + j * LW < nt) epi(wave + j * LW, acc[j]);
        return;
    }
    const float4* W4 = reinterpret_cast<const float4*>(wp) + lane;
    if constexpr (LEAN) {  // six waves per SIMD: the other waves cover the loads; one block of weights in registers, not two
        for (int t0 = wave; t0 < nt; t0 += 4 * LW) {
            f32x4 acc[4];
            unsigned wo[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int t = t0 + j * LW, tc = t < nt ? t : nt - 1;
                const float bv = bias ? bias[tc * 16 + (lane & 15)] : 0.0f;
                acc[j] = f32x4{bv, bv, bv, bv};
                wo[j] = (unsigned)(tc * kg * 64);
            }
            for (int g = 0; g < kg; g++) {
                float4 w[4];
#pragma unroll
                for (int j = 0; j < 4; j++) w[j] = ldg4(W4 + wo[j] + g * 64);
                const float4 x = X4[g * 64];
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = mfma4(x, w[j], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (t0 + j * LW < nt) epi(t0 + j * LW, acc[j]);
        }
        return;
    }
    for (int t0 = wave; t0 < nt; t0 += 4 * LW) {  // any shape: groups of four tiles, weights one block ahead
        f32x4 acc[4];
        const float4* wb[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int t = t0 + j * LW, tc = t < nt ? t : nt - 1;
            const float bv = bias ? bias[tc * 16 + (lane & 15)] : 0.0f;
            acc[j] = f32x4{bv, bv, bv, bv};
            wb[j] = W4 + (size_t)tc * kg * 64;
        }
        float4 w0[4], w1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) w0[j] = ldg4(wb[j]);
        int g = 0;
        for (; g + 1 < kg; g += 2) {
#pragma unroll
            for (int j = 0; j < 4; j++) w1[j] = ldg4(wb[j] + (g + 1) * 64);
            const float4 x0 = X4[g * 64];
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = mfma4(x0, w0[j], acc[j]);
            const int gn = g + 2 < kg ? g + 2 : g + 1;
#pragma unroll
            for (int j = 0; j < 4; j++) w0[j] = ldg4(wb[j] + gn * 64);
            const float4 x1 = X4[(g + 1) * 64];
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = mfma4(x1, w1[j], acc[j]);
        }
        if (g < kg) {
            const float4 x0 = X4[g * 64];
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = mfma4(x0, w0[j], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (t0 + j * LW < nt) epi(t0 + j * LW, acc[j]);
    }
}

// Few output tiles, long reduction (second layers, and the backward pass through a first layer): the reduction blocks of a tile
// are split over LW / nt waves, the partial tiles meet in `red` (LW x 64 float4) and are added in part order.  Every thread of the
// workgroup calls ks_mma (one barrier inside; the epilogue runs after it on waves 0 .. nt-1).
template <bool BIAS, bool F>
__device__ __forceinline__ void ks_load(KsW& W, const float* __restrict__ wp, const float* __restrict__ bias, int nt, int kg, int wave, int lane) {
    W.fast = false;
    W.bv = 0.0f;
    if (!F && 2 * nt > LW) return;
    const int parts = LW / nt;
    if constexpr (BIAS) W.bv = bias[(wave < nt ? wave : 0) * 16 + (lane & 15)];
    W.fast = F;
    if (!W.fast || wave >= nt * parts) return;
    const int t = wave % nt, part = wave / nt;
    const int g0 = part * kg / parts, g1 = (part + 1) * kg / parts;
    const float4* wb = reinterpret_cast<const float4*>(wp) + (size_t)t * kg * 64 + lane;
    // (all KSB, clamped to a block that exists: see wide_load.  With fewer reduction blocks than parts -- 32 planes over 8 parts -- a part's
    // range is EMPTY, g1 - 1 may be -1: round 4's first form read 1 KiB in front of the layer's operands there, never used, but a fault
    // when the operands start their allocation -- found by the randomised plumbing test)
    const int gl = g1 > g0 ? g1 - 1 : (g0 < kg ? g0 : kg - 1);
#pragma unroll
    for (int d = 0; d < KSB; d++) W.w[d] = ldg4(wb + (g0 + d < g1 ? g0 + d : gl) * 64);
}
template <bool F, typename Epi, typename Pf>
__device__ __forceinline__ void ks_mma(const KsW& W, const float* __restrict__ wp, const float* __restrict__ bias, int nt, int kg, const float* Xs,
                                       float* red, int wave, int lane, Epi epi, Pf) {
    const float4* W4 = reinterpret_cast<const float4*>(wp);
    const float4* X4 = reinterpret_cast<const float4*>(Xs) + lane;
    if (!F && 2 * nt > LW) {  // enough tiles for every wave: whole reductions
        for (int t = wave; t < nt; t += LW) {
            const float bv = bias ? bias[t * 16 + (lane & 15)] : 0.0f;
            f32x4 a0 = {bv, bv, bv, bv}, a1 = {0.0f, 0.0f, 0.0f, 0.0f};
            const float4* wb = W4 + (size_t)t * kg * 64 + lane;
            int g = 0;
            for (; g + 1 < kg; g += 2) {
                const float4 wa = ldg4(wb + g * 64), wc = ldg4(wb + (g + 1) * 64);
                a0 = mfma4(X4[g * 64], wa, a0);
                a1 = mfma4(X4[(g + 1) * 64], wc, a1);
            }
            if (g < kg) a0 = mfma4(X4[g * 64], ldg4(wb + g * 64), a0);
            epi(t, a0 + a1);
        }
        __syncthreads();
        return;
    }
    const int parts = LW / nt;
    if (wave < nt * parts) {
        const int t = wave % nt, part = wave / nt;
        const int g0 = part * kg / parts, g1 = (part + 1) * kg / parts;
        f32x4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (F || W.fast) {
#pragma unroll
            for (int d = 0; d < KSB; d += 2) {
                if (g0 + d < g1) a0 = mfma4(X4[(g0 + d) * 64], W.w[d], a0);
                if (g0 + d + 1 < g1) a1 = mfma4(X4[(g0 + d + 1) * 64], W.w[d + 1], a1);
            }
        } else {
            const float4* wb = W4 + (size_t)t * kg * 64 + lane;
            int g = g0;
            for (; g + 3 < g1; g += 4) {  // four blocks of weights in flight
                const float4 wa = ldg4(wb + g * 64), wc = ldg4(wb + (g + 1) * 64), wd = ldg4(wb + (g + 2) * 64), we = ldg4(wb + (g + 3) * 64);
                a0 = mfma4(X4[g * 64], wa, a0);
                a1 = mfma4(X4[(g + 1) * 64], wc, a1);
                a0 = mfma4(X4[(g + 2) * 64], wd, a0);
                a1 = mfma4(X4[(g + 3) * 64], we, a1);
            }
            for (; g < g1; g++) a0 = mfma4(X4[g * 64], ldg4(wb + g * 64), a0);
        }
        const f32x4 s = a0 + a1;
        reinterpret_cast<float4*>(red)[(part * nt + t) * 64 + lane] = make_float4(s[0], s[1], s[2], s[3]);
    }
    __syncthreads();
    if (wave < nt) {
        f32x4 s = {W.bv, W.bv, W.bv, W.bv};
        for (int p = 0; p < parts; p++) {
            const float4 v = reinterpret_cast<const float4*>(red)[(p * nt + wave) * 64 + lane];
            s = s + f32x4{v.x, v.y, v.z, v.w};
        }
        epi(wave, s);
    }
}

// a T-layout accumulator tile -> the packed LDS copy the next layer multiplies (four ds_write_b32)
__device__ __forceinline__ void lds_put_T(float* dst, int t, int lane, const f32x4& v) {
    const int f = lane & 15, sq = lane >> 4;
    float* p = dst + (((t * 64 + (f >> 2) * 16 + 4 * sq) << 2) + (f & 3));
    p[0] = v[0];
    p[4] = v[1];
    p[8] = v[2];
    p[12] = v[3];
}
__device__ __forceinline__ void g_put_T(float* blocks, int t, int lane, const f32x4& v) {
    reinterpret_cast<float4*>(blocks)[t * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ f32x4 g_get_T(const float* blocks, int t, int lane) {
    const float4 v = reinterpret_cast<const float4*>(blocks)[t * 64 + lane];
    return f32x4{v.x, v.y, v.z, v.w};
}
// a packed LDS buffer of nt feature tiles -> T blocks in global memory (four ds_read_b32 + one 16-byte store per lane and tile)
__device__ __forceinline__ void save_T_from_pk(const float* lds, int nt, float* blocks, int tid) {
    for (int u = tid; u < nt * 64; u += LT) {
        const int t = u >> 6, lane = u & 63, f = lane & 15, sq = lane >> 4;
        const float* p = lds + (((t * 64 + (f >> 2) * 16 + 4 * sq) << 2) + (f & 3));
        reinterpret_cast<float4*>(blocks)[u] = make_float4(p[0], p[4], p[8], p[12]);
    }
}
__device__ __forceinline__ void copy_f4(float* dst, const float* src, int nfloat4, int tid) {
    for (int i = tid; i < nfloat4; i += LT) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
}

// ------------------------------------------------------------------------------------------------------------------
// row phases: RPS = 32 threads per sample (e = tid >> 5, j = tid & 31); reductions stay inside a 32-lane half wave
// ------------------------------------------------------------------------------------------------------------------
// (four DPP row rotations + ONE ds_bpermute instead of five ds_bpermute round trips: the loss rows are a chain of such reductions on
// waves that have nothing else to issue meanwhile)
__device__ __forceinline__ float half_sum(float v) {
    v = mz::butterfly16(v);
    return v + __shfl_xor(v, 16, 64);
}
__device__ __forceinline__ float half_max(float v) {
    v = mz::butterfly16_max(v);
    const float o = __shfl_xor(v, 16, 64);
    return o > v ? o : v;
}
__device__ __forceinline__ float half_min(float v) {
    v = mz::butterfly16_min(v);
    const float o = __shfl_xor(v, 16, 64);
    return o < v ? o : v;
}

// normalize_hidden_state, util.py:31-36, over the H real features of each sample; padding features stay zero
__device__ __forceinline__ void normalize_fwd(const float* HN, float* HS, int H, int h_t, int tid) {
    const int e = tid >> 5, j = tid & 31;
    float mn = __uint_as_float(0x7f800000u), mx = __uint_as_float(0xff800000u);
    for (int k = j; k < H; k += 32) {
        const float v = HN[pk(k, e)];
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
    mx = half_max(mx);
    mn = half_min(mn);
    const float d = (mx - mn) + 1e-8f;
    for (int k = j; k < h_t * 16; k += 32) HS[pk(k, e)] = k < H ? (HN[pk(k, e)] - mn) / d : 0.0f;
}

// backward of the same: G = dL/dh (scaled by gs), U = the un-normalised state; DU = dL/du (+ R).  The min / max terms go to the
// FIRST index that attains them (autograd's min / max over a dimension route the gradient to the returned index).
__device__ __forceinline__ void normalize_bwd(const float* G, float gs, const float* U, const float* R, float* DU, int H, int h_t, int tid) {
    const int e = tid >> 5, j = tid & 31;
    float mn = __uint_as_float(0x7f800000u), mx = __uint_as_float(0xff800000u);
    float imn = 1e9f, imx = 1e9f;  // first index that attains the bound, as a float (indices are small integers: exact)
    for (int k = j; k < H; k += 32) {
        const float v = U[pk(k, e)];
        if (v < mn) { mn = v; imn = (float)k; }
        if (v > mx) { mx = v; imx = (float)k; }
    }
    const float gmn = half_min(mn), gmx = half_max(mx);
    imn = half_min(mn == gmn ? imn : 1e9f);
    imx = half_min(mx == gmx ? imx : 1e9f);
    mn = gmn;
    mx = gmx;
    const float d = (mx - mn) + 1e-8f;
    float s1 = 0.0f, s2 = 0.0f;
    for (int k = j; k < H; k += 32) {
        const float g = G[pk(k, e)] * gs, h = (U[pk(k, e)] - mn) / d;
        s1 += g;
        s2 += g * h;
    }
    s1 = half_sum(s1);
    s2 = half_sum(s2);
    const float dmn = (s2 - s1) / d, dmx = -s2 / d;
    for (int k = j; k < h_t * 16; k += 32) {
        float v = 0.0f;
        if (k < H) {
            v = G[pk(k, e)] * gs / d;
            if ((float)k == imn) v += dmn;
            if ((float)k == imx) v += dmx;
            if (R) v += R[pk(k, e)];
        }
        DU[pk(k, e)] = v;
    }
}

__device__ __forceinline__ float signed_hyperbolic(float x) {  // util.py:20-22
    const float sg = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
    return sg * (sqrtf(fabsf(x) + 1.0f) - 1.0f) + 0.001f * x;
}

// scalar -> the two bins of its categorical target (util.py:48-59 through :96-116), float32 in the reference's op order
struct TwoHot { int lo, hi; float plo, phi; };
__device__ __forceinline__ TwoHot two_hot(float scalar, int S) {
    const float half = (float)((S - 1) / 2), mnv = -half, span = half - mnv;
    float x = signed_hyperbolic(scalar);
    x = x < mnv ? mnv : (x > half ? half : x);
    const float bin = (x - mnv) / span * (float)(S - 1);
    const float lo = floorf(bin), hi = ceilf(bin);
    const float lov = (lo / ((float)S - 1.0f)) * span + mnv, hiv = (hi / ((float)S - 1.0f)) * span + mnv;
    TwoHot t;
    t.plo = (hiv - x) / (hiv - lov + 1e-5f);
    t.phi = 1.0f - t.plo;
    t.lo = (int)lo;
    t.hi = (int)hi;
    return t;
}

// One head's loss rows for the 16 samples of the tile.  LG: logits rows [16][lgs]; kind 0: soft target vector `pi` (policy),
// 1: two-hot target of `tgt` (categorical value / reward), 2: squared error (support size 1).  Writes dL/dlogits (already scaled by
// w / (B K), pipeline.py:597-600) to DL (pk order, padded features zero) and returns this thread's share of sum_e w_e loss_e; if
// prio != nullptr also the step-0 priorities |value - target| (pipeline.py:603-609).
// pi_first = pi_row[j] (thread j's first element), requested by the caller at kernel start.
__device__ __forceinline__ float head_loss_rows(const float* LG, int lgs, int S, int s_t, int kind, const float* pi_row, float pi_first, float tgt, float w,
                                                float scale, bool valid, float* DL, float* prio_out, int tid) {
    const int e = tid >> 5, j = tid & 31;
    const float* z = LG + e * lgs;
    float loss = 0.0f;
    if (kind == 2) {
        const float d = z[0] - tgt;
        for (int c = j; c < s_t * 16; c += 32) DL[pk(c, e)] = (c == 0 && valid) ? 2.0f * d * scale : 0.0f;
        if (j == 0 && valid) {
            loss = w * d * d;
            if (prio_out) *prio_out = fabsf(z[0] - tgt);
        }
        return loss;
    }
    TwoHot th = {0, 0, 0.0f, 0.0f};
    if (kind == 1) th = two_hot(tgt, S);
    float m = __uint_as_float(0xff800000u);
    for (int c = j; c < S; c += 32) m = z[c] > m ? z[c] : m;
    m = half_max(m);
    float se = 0.0f, st = 0.0f, dot = 0.0f, ev = 0.0f;
    const int half = (S - 1) / 2;
    for (int c = j; c < S; c += 32) {
        const float ex = mz::expf_det(z[c] - m);
        se += ex;
        const float t = kind == 0 ? (c == j ? pi_first : pi_row[c]) : ((c == th.lo ? th.plo : 0.0f) + (c == th.hi ? th.phi : 0.0f));
        st += t;
        dot += t * (z[c] - m);
        ev += ex * (float)(c - half);
    }
    se = half_sum(se);
    st = half_sum(st);
    dot = half_sum(dot);
    const float lse = logf(se);
    for (int c = j; c < s_t * 16; c += 32) {
        float v = 0.0f;
        if (c < S && valid) {
            const float p = mz::expf_det(z[c] - m) / se;
            const float t = kind == 0 ? (c == j ? pi_first : pi_row[c]) : ((c == th.lo ? th.plo : 0.0f) + (c == th.hi ? th.phi : 0.0f));
            v = (p * st - t) * scale;
        }
        DL[pk(c, e)] = v;
    }
    if (prio_out) {  // logits_to_transformed_expected_value, util.py:70-93
        ev = half_sum(ev);
        if (j == 0 && valid) *prio_out = fabsf(mz::signed_parabolic(ev / se) - tgt);
    }
    if (j == 0 && valid) loss = w * (st * lse - dot);
    return loss;
}

// sum of `v` over the workgroup -> out (thread 0 writes); scratch: LW floats of LDS
__device__ __forceinline__ void block_sum_store(float v, float* scratch, float* out, int tid) {
    v = mz::butterfly16(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        float s = 0.0f;
        for (int i = 0; i < LW; i++) s += scratch[i];
        *out = s;
    }
}

// LDS carve-out shared by the stage kernels (float offsets)
struct LLds {
    int X, H1, DZ, HN, HS, G, R, LG, DL, RED, MISC, total;
};

__device__ __forceinline__ size_t blk(int step, int tiles, int tile, int ft) { return ((size_t)(step * tiles + tile) * ft) * 256; }

__device__ __forceinline__ int sample_row(const LBatch& b, int s) { return (int)b.idx[s < b.B ? s : b.B - 1]; }  // (clamped, never guarded: see LOAD ORDER)
__device__ __forceinline__ int sample_action(const LBatch& b, int row, int K, int k) {
    return b.action_bytes == 1 ? (int)reinterpret_cast<const int8_t*>(b.action)[(size_t)row * K + k]
                               : (int)reinterpret_cast<const int16_t*>(b.action)[(size_t)row * K + k];
}

// ------------------------------------------------------------------------------------------------------------------
// stage kernels
// ------------------------------------------------------------------------------------------------------------------
#define MZL_STAMP(slot) do { if (sv.stamps && blockIdx.x == 0 && threadIdx.x == 0) sv.stamps[slot] = (long long)__builtin_readcyclecounter(); } while (0)

// LOAD ORDER inside the stage kernels.  vmcnt retires in order: waiting for a load waits for every load issued before it.  So the
// kernels request their global operands in the order of first USE -- chain tensors and row indices first, then the first GEMM's
// weights, then whatever the loss rows need -- and each later GEMM's weights one phase ahead (before the previous GEMM's MFMAs).
// A dependent load (row index -> target) is issued as soon as its address is there, never behind a prefetch it would drain.

// one float4 per thread of a chain tensor (h_t * 64 float4s <= LT in the register form; larger states take the loop)
__device__ __forceinline__ float4 chain_ld(const float* src, int n4, int tid) { return reinterpret_cast<const float4*>(src)[tid < n4 ? tid : n4 - 1]; }
template <bool F>
__device__ __forceinline__ void chain_st(float* lds_dst, const float* src, float4 v, int n4, int tid) {
    if (tid < n4) reinterpret_cast<float4*>(lds_dst)[tid] = v;
    if (!F)
        for (int i = tid + LT; i < n4; i += LT) reinterpret_cast<float4*>(lds_dst)[i] = reinterpret_cast<const float4*>(src)[i];
}

// representation net forward (network.py:151-156 + normalisation :116): observations gathered from the replay ring
template <bool F>
__global__ __launch_bounds__(LT) void k_learn_repr(LNet net, LSave sv, LBatch bt, LLds o) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), tile = blockIdx.x, tiles = bt.tiles;
    const int e = tid >> 5, j = tid & 31, s = tile * TILE + e;
    const bool valid = s < bt.B;
    const int row = sample_row(bt, s);
    WideW w1;
    KsW w2;
    wide_load<6, true, F>(w1, net.L[REP0].wp, net.L[REP0].b, net.p_t, net.in_t, wave, lane);
    // X <- observations (pk order), zero padded; samples past the batch read as zeros
    for (int k = j; k < net.in_t * 16; k += 32) {
        float v = 0.0f;
        if (valid && k < net.in_dim)
            v = bt.state_i8 ? (float)reinterpret_cast<const int8_t*>(bt.state)[(size_t)row * net.in_dim + k]
                            : reinterpret_cast<const float*>(bt.state)[(size_t)row * net.in_dim + k];
        lds[o.X + pk(k, e)] = v;
    }
    if (j < net.K) sv.actc[(j * tiles + tile) * TILE + e] = valid ? sample_action(bt, row, net.K, j) : -1;  // (K <= 32)
    __syncthreads();
    ks_load<true, F>(w2, net.L[REP1].wp, net.L[REP1].b, net.h_t, net.p_t, wave, lane);
    save_T_from_pk(lds + o.X, net.in_t, sv.in_rep + blk(0, tiles, tile, net.in_t), tid);
    float* h1b = sv.h1_rep + blk(0, tiles, tile, net.p_t);
    wide_mma<F, 6>(w1, net.L[REP0].wp, net.L[REP0].b, net.p_t, net.in_t, lds + o.X, wave, lane, [&](int t, f32x4 a) {
        const f32x4 r = {fmaxf(a[0], 0.0f), fmaxf(a[1], 0.0f), fmaxf(a[2], 0.0f), fmaxf(a[3], 0.0f)};
        g_put_T(h1b, t, lane, r);
        lds_put_T(lds + o.H1, t, lane, r);
    }, NoPf{});
    __syncthreads();
    ks_mma<F>(w2, net.L[REP1].wp, net.L[REP1].b, net.h_t, net.p_t, lds + o.H1, lds + o.RED, wave, lane,
           [&](int t, f32x4 a) { lds_put_T(lds + o.HN, t, lane, a); }, NoPf{});
    __syncthreads();
    normalize_fwd(lds + o.HN, lds + o.HS, net.H, net.h_t, tid);
    __syncthreads();
    const size_t cb = blk(0, tiles, tile, net.h_t);
    copy_f4(sv.uc + cb, lds + o.HN, net.h_t * 64, tid);
    copy_f4(sv.hc + cb, lds + o.HS, net.h_t * 64, tid);
}

// one unroll step of the forward sweep (pipeline.py:579-592); see the header for the roles
// OCC: workgroups per CU the build is compiled for (launch bound: OCC x LW / 4 waves per SIMD).  3 (streaming form, heads only): the heads need 53 KB of LDS (LLds without the
// dynamics role's blocks, learner.hip) and 80 VGPRs -- a third co-resident workgroup's MFMAs under the others' barriers and loss rows
template <bool F, int OCC = 1>
__global__ __launch_bounds__(LT, OCC * LW / 4) void k_learn_unroll(LNet net, LSave sv, LBatch bt, LLds o, int k0, int role0) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), tile = blockIdx.x, tiles = bt.tiles;
    // grid (tiles, roles, steps): role = role0 + blockIdx.y; the launch's step index blockIdx.z addresses h_k for the dynamics / policy / value
    // roles and u_k (reward of step k - 1) for the reward role, whose steps are therefore shifted by one
    // (the all-heads launch is grid.y == 3: its reward role is shifted by one step.  Round 4's first form tested gridDim.z > 1 -- "more than
    // one step in the launch" -- which dropped the reward head altogether for unroll_steps == 1: found by the randomised learner test)
    const int role = role0 + (int)blockIdx.y, k = k0 + (int)blockIdx.z + ((role == 3 && gridDim.y > 1) ? 1 : 0);
    const int K = net.K, hf = net.h_t * 64;
    const int e = tid >> 5, j = tid & 31, s = tile * TILE + e;
    WideW w1;
    KsW w2;
    if (OCC != 3 && role == 0) {  // (the heads-only build carries no dynamics code)
        if (k >= K) return;
        MZL_STAMP(0);
        const float* hsrc = sv.hc + blk(k, tiles, tile, net.h_t);
        const float4 xv = chain_ld(hsrc, hf, tid);
        const int act = sv.actc[(k * tiles + tile) * TILE + e];
        wide_load<6, true, F>(w1, net.L[DYN0].wp, net.L[DYN0].b, net.p_t, net.h_t + net.a_t, wave, lane);
        // X <- [h_k | onehot(a_k)] (network.py:191-193)
        chain_st<F>(lds + o.X, hsrc, xv, hf, tid);
        for (int a = j; a < net.a_t * 16; a += 32) lds[o.X + pk(net.h_t * 16 + a, e)] = a == act ? 1.0f : 0.0f;
        __syncthreads();
        MZL_STAMP(1);
        ks_load<true, F>(w2, net.L[DYN1].wp, net.L[DYN1].b, net.h_t, net.p_t, wave, lane);
        save_T_from_pk(lds + o.X, net.h_t + net.a_t, sv.x + blk(k, tiles, tile, net.h_t + net.a_t), tid);
        float* h1b = sv.h1_dyn + blk(k, tiles, tile, net.p_t);
        wide_mma<F, 6>(w1, net.L[DYN0].wp, net.L[DYN0].b, net.p_t, net.h_t + net.a_t, lds + o.X, wave, lane, [&](int t, f32x4 a) {
            const f32x4 r = {fmaxf(a[0], 0.0f), fmaxf(a[1], 0.0f), fmaxf(a[2], 0.0f), fmaxf(a[3], 0.0f)};
            g_put_T(h1b, t, lane, r);
            lds_put_T(lds + o.H1, t, lane, r);
        }, NoPf{});
        __syncthreads();
        MZL_STAMP(2);
        ks_mma<F>(w2, net.L[DYN1].wp, net.L[DYN1].b, net.h_t, net.p_t, lds + o.H1, lds + o.RED, wave, lane,
               [&](int t, f32x4 a) { lds_put_T(lds + o.HN, t, lane, a); }, NoPf{});
        __syncthreads();
        MZL_STAMP(3);
        normalize_fwd(lds + o.HN, lds + o.HS, net.H, net.h_t, tid);
        __syncthreads();
        MZL_STAMP(4);
        const size_t cb = blk(k + 1, tiles, tile, net.h_t);
        copy_f4(sv.uc + cb, lds + o.HN, hf, tid);
        copy_f4(sv.hc + cb, lds + o.HS, hf, tid);
        MZL_STAMP(5);
        return;
    }
    // heads: policy_k, value_k on h_k; reward_{k-1} on u_k (network.py:195-196 reads the un-normalised state)
    const int step = role == 3 ? k - 1 : k;
    if (step < 0 || step >= K) return;
    const int l0 = role == 1 ? POL0 : (role == 2 ? VAL0 : REW0), l1 = l0 + 1;
    const int S = role == 1 ? net.A : (role == 2 ? net.Sv : net.Sr), s_t = role == 1 ? net.a_t : (role == 2 ? net.sv_t : net.sr_t);
    const int sb = 16 * role;
    MZL_STAMP(sb + 0);
    const bool valid = s < bt.B;
    const int row = sample_row(bt, s);  // (first: the targets' addresses hang on it)
    const float* xsrc = (role == 3 ? sv.uc : sv.hc) + blk(k, tiles, tile, net.h_t);
    const float4 xv = chain_ld(xsrc, hf, tid);
    wide_load<4, true, F>(w1, net.L[l0].wp, net.L[l0].b, net.p_t, net.h_t, wave, lane);
    // this sample's targets: used in the loss rows, two GEMMs from here
    const float wl = bt.w[s < bt.B ? s : bt.B - 1], w = valid ? wl : 0.0f;
    const float tgt = (role == 2 ? bt.value : bt.reward)[(size_t)row * K + step];
    const float* pi_row = bt.pi + ((size_t)row * K + step) * net.A;
    const float pi_first = pi_row[j < net.A ? j : 0];
    float* h1_all = role == 1 ? sv.h1_pol : (role == 2 ? sv.h1_val : sv.h1_rew);
    float* dz0_all = role == 1 ? sv.dz_pol0 : (role == 2 ? sv.dz_val0 : sv.dz_rew0);
    float* dz1_all = role == 1 ? sv.dz_pol1 : (role == 2 ? sv.dz_val1 : sv.dz_rew1);
    float* dx_all = role == 1 ? sv.dxp : (role == 2 ? sv.dxv : sv.dxr);
    chain_st<F>(lds + o.X, xsrc, xv, hf, tid);
    __syncthreads();
    MZL_STAMP(sb + 1);
    ks_load<true, F>(w2, net.L[l1].wp, net.L[l1].b, s_t, net.p_t, wave, lane);
    MZL_STAMP(sb + 8);
    if (role == 3) save_T_from_pk(lds + o.X, net.h_t, sv.u_in + blk(step, tiles, tile, net.h_t), tid);
    MZL_STAMP(sb + 9);
    float* h1b = h1_all + blk(step, tiles, tile, net.p_t);
    f32x4 h1k[4];  // this wave's relu outputs (its tiles wave + j LW): the backward pass's gate, same lanes (register form)
#pragma unroll
    for (int i = 0; i < 4; i++) h1k[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const bool keep = F || w1.fast;
    wide_mma<F, 4, OCC == 3>(w1, net.L[l0].wp, net.L[l0].b, net.p_t, net.h_t, lds + o.X, wave, lane, [&](int t, f32x4 a) {
        const f32x4 r = {fmaxf(a[0], 0.0f), fmaxf(a[1], 0.0f), fmaxf(a[2], 0.0f), fmaxf(a[3], 0.0f)};
        g_put_T(h1b, t, lane, r);
        lds_put_T(lds + o.H1, t, lane, r);
        if (keep) {
            const int jj = (t - wave) / LW;
            if (jj == 0) h1k[0] = r;
            if (jj == 1) h1k[1] = r;
            if (jj == 2) h1k[2] = r;
            if (jj == 3) h1k[3] = r;
        }
    }, NoPf{});
    MZL_STAMP(sb + 10);
    __syncthreads();
    MZL_STAMP(sb + 2);
    if constexpr (OCC != 3) wide_load<2, false, F>(w1, net.L[l1].wtp, nullptr, net.p_t, s_t, wave, lane);  // the backward pass's operands, one phase ahead
    ks_mma<F>(w2, net.L[l1].wp, net.L[l1].b, s_t, net.p_t, lds + o.H1, lds + o.RED, wave, lane, [&](int t, f32x4 a) {
        const int f = lane & 15, sq = lane >> 4;
#pragma unroll
        for (int i = 0; i < 4; i++) lds[o.LG + (4 * sq + i) * net.lgs + 16 * t + f] = a[i];
    }, NoPf{});
    __syncthreads();
    MZL_STAMP(sb + 3);
    if constexpr (OCC == 3) wide_load<2, false, true>(w1, net.L[l1].wtp, nullptr, net.p_t, s_t, wave, lane);  // (80-VGPR build: behind the streamed GEMM, under the
                                                                                                          // loss rows; register form: the launcher checks p_t <= 4 LW, s_t <= 2)
    ks_load<false, F>(w2, net.L[l0].wtp, nullptr, net.h_t, net.p_t, wave, lane);
    {
        const float scale = w / ((float)bt.B * (float)K);
        const int kind = role == 1 ? 0 : (S == 1 ? 2 : 1);
        float* prio = (role == 2 && step == 0 && valid) ? bt.prio + s : nullptr;
        const float part = head_loss_rows(lds + o.LG, net.lgs, S, s_t, kind, pi_row, pi_first, tgt, w, scale, valid, lds + o.DL, prio, tid);
        block_sum_store(part, lds + o.MISC, sv.lossp + (size_t)(step * 3 + (role - 1)) * tiles + tile, tid);  // (one barrier inside)
    }
    __syncthreads();
    MZL_STAMP(sb + 4);
    save_T_from_pk(lds + o.DL, s_t, dz1_all + blk(step, tiles, tile, s_t), tid);
    float* dzb = dz0_all + blk(step, tiles, tile, net.p_t);
    const bool kept = F || (keep && w1.fast);
    wide_mma<F || OCC == 3, 2>(w1, net.L[l1].wtp, nullptr, net.p_t, s_t, lds + o.DL, wave, lane, [&](int t, f32x4 a) {
        f32x4 h;
        if (kept) {
            const int jj = (t - wave) / LW;
            h = jj == 0 ? h1k[0] : (jj == 1 ? h1k[1] : (jj == 2 ? h1k[2] : h1k[3]));
        } else {
            h = g_get_T(h1b, t, lane);  // (this lane's own store above: relu'(z) = [relu(z) > 0])
        }
        const f32x4 r = {h[0] > 0.0f ? a[0] : 0.0f, h[1] > 0.0f ? a[1] : 0.0f, h[2] > 0.0f ? a[2] : 0.0f, h[3] > 0.0f ? a[3] : 0.0f};
        g_put_T(dzb, t, lane, r);
        lds_put_T(lds + o.DZ, t, lane, r);
    }, NoPf{});
    __syncthreads();
    MZL_STAMP(sb + 5);
    ks_mma<F>(w2, net.L[l0].wtp, nullptr, net.h_t, net.p_t, lds + o.DZ, lds + o.RED, wave, lane,
           [&](int t, f32x4 a) { lds_put_T(lds + o.G, t, lane, a); }, NoPf{});
    __syncthreads();
    MZL_STAMP(sb + 6);
    copy_f4(dx_all + blk(step, tiles, tile, net.h_t), lds + o.G, hf, tid);
    MZL_STAMP(sb + 7);
}

// The forward chain for SMALL batches, cut across the planes like k_learn_back_sliced: workgroup (tile, part) computes its slice of the
// first layer (one plane tile per wave) and a PARTIAL second layer over that slice; the consumer -- the next stage's workgroups, each
// for itself -- adds the np partials and the bias, normalises, and goes on.  k == -1: representation net on the observations -> partial
// u_0; 0 <= k < K: dynamics_k -> partial u_{k+1}; k == K (grid (tiles, 1)): only finishes u_K / h_K for the heads and the backward sweep.
// Part 0 of every stage also writes the finished u_k / h_k (uc / hc) and the T blocks that are not sliced (x_k, the observation).
template <bool F>
__global__ __launch_bounds__(LT) void k_learn_fwd_sliced(LNet net, LSave sv, LBatch bt, LLds o, int k, int np) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), tile = blockIdx.x, tiles = bt.tiles;
    const int part = blockIdx.y, K = net.K, hf = net.h_t * 64;
    const int e = tid >> 5, jr = tid & 31, smp = tile * TILE + e;
    const int l0 = k < 0 ? REP0 : DYN0, l1 = l0 + 1, kg1 = k < 0 ? net.in_t : net.h_t + net.a_t;
    const int tb = part * net.p_t / np, te = (part + 1) * net.p_t / np;
    const int ci = tid < hf ? tid : hf - 1;
    // ---- requests in order of use: the state's partials (+ bias), the action, then both layers' operands of this wave ----
    float4 us = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    int act = -1;
    const int kk = k < 0 ? 0 : k;
    {
        const float* ub = sv.up + blk(kk, tiles, tile, net.h_t) * DX_PARTS;
#pragma unroll
        for (int pp = 0; pp < DX_PARTS; pp++) {
            const float4 v = reinterpret_cast<const float4*>(ub + (size_t)(pp < np ? pp : 0) * net.h_t * 256)[ci];
            const float m = pp < np ? 1.0f : 0.0f;
            us = make_float4(us.x + m * v.x, us.y + m * v.y, us.z + m * v.z, us.w + m * v.w);
        }
        const float* bsrc = (k <= 0 ? net.L[REP1].b : net.L[DYN1].b) + (ci >> 6) * 16 + ((ci >> 4) & 3) * 4;  // the four features of this float4
        const float4 bv = *reinterpret_cast<const float4*>(bsrc);
        us = make_float4(us.x + bv.x, us.y + bv.y, us.z + bv.z, us.w + bv.w);
        act = sv.actc[((kk < K ? kk : K - 1) * tiles + tile) * TILE + e];
    }
    const int row = sample_row(bt, smp);
    const int t = tb + wave < te ? tb + wave : te - 1;
    const bool t_ok = tb + wave < te && k < K;
    const float4* W1 = reinterpret_cast<const float4*>(net.L[l0].wp) + lane;
    float4 w1[WKG];
#pragma unroll
    for (int d = 0; d < WKG; d++) w1[d] = ldg4(W1 + ((size_t)t * kg1 + (d < kg1 ? d : kg1 - 1)) * 64);
    const float b1 = net.L[l0].b[t * 16 + (lane & 15)];
    const int kparts = LW / net.h_t, ot = wave % net.h_t, kp = wave / net.h_t;
    const int nb = (te - tb) / kparts, g0 = tb + kp * nb;
    const float4* W2 = reinterpret_cast<const float4*>(net.L[l1].wp) + ((size_t)ot * net.p_t) * 64 + lane;
    float4 w2[4];
#pragma unroll
    for (int d = 0; d < 4; d++) w2[d] = ldg4(W2 + (size_t)(g0 + (d < nb ? d : nb - 1)) * 64);
    // ---- the layer input ----
    if (k < 0) {
        const bool valid = smp < bt.B;
        for (int f = jr; f < net.in_t * 16; f += 32) {
            float v = 0.0f;
            if (valid && f < net.in_dim)
                v = bt.state_i8 ? (float)reinterpret_cast<const int8_t*>(bt.state)[(size_t)row * net.in_dim + f]
                                : reinterpret_cast<const float*>(bt.state)[(size_t)row * net.in_dim + f];
            lds[o.X + pk(f, e)] = v;
        }
        if (part == 0 && jr < K) sv.actc[(jr * tiles + tile) * TILE + e] = valid ? sample_action(bt, row, K, jr) : -1;
        __syncthreads();
        if (part == 0) save_T_from_pk(lds + o.X, net.in_t, sv.in_rep + blk(0, tiles, tile, net.in_t), tid);
    } else {
        if (tid < hf) reinterpret_cast<float4*>(lds + o.HN)[tid] = us;
        __syncthreads();
        normalize_fwd(lds + o.HN, lds + o.X, net.H, net.h_t, tid);  // h_k straight into the layer-input buffer
        for (int a = jr; a < net.a_t * 16; a += 32) lds[o.X + pk(net.h_t * 16 + a, e)] = a == act ? 1.0f : 0.0f;
        __syncthreads();
        if (part == 0) {
            const size_t cb = blk(k, tiles, tile, net.h_t);
            copy_f4(sv.uc + cb, lds + o.HN, hf, tid);
            copy_f4(sv.hc + cb, lds + o.X, hf, tid);
            if (k < K) save_T_from_pk(lds + o.X, net.h_t + net.a_t, sv.x + blk(k, tiles, tile, net.h_t + net.a_t), tid);
        }
        if (k >= K) return;
    }
    // ---- first layer: this wave's plane tile ----
    float* h1b = (k < 0 ? sv.h1_rep : sv.h1_dyn) + blk(kk, tiles, tile, net.p_t);
    if (t_ok) {
        const float4* X4 = reinterpret_cast<const float4*>(lds + o.X) + lane;
        f32x4 a = {b1, b1, b1, b1};
#pragma unroll
        for (int d = 0; d < WKG; d++)
            if (d < kg1) a = mfma4(X4[d * 64], w1[d], a);
        const f32x4 r = {fmaxf(a[0], 0.0f), fmaxf(a[1], 0.0f), fmaxf(a[2], 0.0f), fmaxf(a[3], 0.0f)};
        g_put_T(h1b, t, lane, r);
        lds_put_T(lds + o.H1, t, lane, r);
    }
    __syncthreads();
    // ---- second layer over the slice: partial u ----
    {
        const float4* X4 = reinterpret_cast<const float4*>(lds + o.H1) + lane;
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int d = 0; d < 4; d++)
            if (d < nb) a = mfma4(X4[(g0 + d) * 64], w2[d], a);
        reinterpret_cast<float4*>(lds + o.RED)[(kp * net.h_t + ot) * 64 + lane] = make_float4(a[0], a[1], a[2], a[3]);
    }
    __syncthreads();
    if (wave < net.h_t) {
        f32x4 sacc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int q = 0; q < kparts; q++) {
            const float4 v = reinterpret_cast<const float4*>(lds + o.RED)[(q * net.h_t + wave) * 64 + lane];
            sacc = sacc + f32x4{v.x, v.y, v.z, v.w};
        }
        lds_put_T(lds + o.G, wave, lane, sacc);
    }
    __syncthreads();
    copy_f4(sv.up + blk(k + 1, tiles, tile, net.h_t) * DX_PARTS + (size_t)part * net.h_t * 256, lds + o.G, hf, tid);
}

// backward sweep, one step of the dynamics chain: k in [0, K) -> dynamics_k; k == -1 -> the representation net
template <bool F>
__global__ __launch_bounds__(LT) void k_learn_back(LNet net, LSave sv, LBatch bt, LLds o, int k) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), tile = blockIdx.x, tiles = bt.tiles;
    const int K = net.K, j = k + 1;  // the state this stage's net produced: u_j / h_j
    const int hf = net.h_t * 64;     // float4s of a chain tensor
    const int l0 = k >= 0 ? DYN0 : REP0, l1 = l0 + 1;
    const int st = k >= 0 ? k : 0;
    // chain tensors first (used at once), then the two GEMMs' operands and the relu gate of this wave's tiles
    const size_t cj = blk(j, tiles, tile, net.h_t);
    const int ci = tid < hf ? tid : hf - 1;
    const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const bool cons = j < K;  // consumers of h_j: dynamics_j, policy_j, value_j (none for j == K)
    const size_t cjc = cons ? cj : 0;
    const float4 ga = reinterpret_cast<const float4*>(sv.dxd + cjc * DX_PARTS)[ci], gb = reinterpret_cast<const float4*>(sv.dxp + cjc)[ci],
                 gc = reinterpret_cast<const float4*>(sv.dxv + cjc)[ci];
    const float4 uu = reinterpret_cast<const float4*>(sv.uc + cj)[ci];
    const float4 rr = reinterpret_cast<const float4*>(sv.dxr + blk(st, tiles, tile, net.h_t))[ci];
    WideW w1;
    KsW w2;
    wide_load<4, false, F>(w1, net.L[l1].wtp, nullptr, net.p_t, net.h_t, wave, lane);
    const float* h1b = (k >= 0 ? sv.h1_dyn : sv.h1_rep) + blk(st, tiles, tile, net.p_t);
    f32x4 gate[4];
    const bool pre = F || w1.fast;
#pragma unroll
    for (int i = 0; i < 4; i++) gate[i] = g_get_T(h1b, wave + i * LW < net.p_t ? wave + i * LW : net.p_t - 1, lane);
    ks_load<false, F>(w2, net.L[DYN0].wtp, nullptr, net.h_t, net.p_t, wave, lane);  // (unused by the representation stage; a branch here would cost a wait)
    if (tid < hf) {
        reinterpret_cast<float4*>(lds + o.G)[tid] = cons ? make_float4((ga.x + gb.x) + gc.x, (ga.y + gb.y) + gc.y, (ga.z + gb.z) + gc.z, (ga.w + gb.w) + gc.w) : z4;
        reinterpret_cast<float4*>(lds + o.HN)[tid] = uu;
        reinterpret_cast<float4*>(lds + o.R)[tid] = rr;
    }
    for (int i = tid + LT; !F && i < hf; i += LT) {  // (states wider than 128 features)
        float4 g = z4;
        if (cons) {
            const float4 a = reinterpret_cast<const float4*>(sv.dxd + cj * DX_PARTS)[i], b = reinterpret_cast<const float4*>(sv.dxp + cj)[i],
                         c = reinterpret_cast<const float4*>(sv.dxv + cj)[i];
            g = make_float4((a.x + b.x) + c.x, (a.y + b.y) + c.y, (a.z + b.z) + c.z, (a.w + b.w) + c.w);
        }
        reinterpret_cast<float4*>(lds + o.G)[i] = g;
        reinterpret_cast<float4*>(lds + o.HN)[i] = reinterpret_cast<const float4*>(sv.uc + cj)[i];
        reinterpret_cast<float4*>(lds + o.R)[i] = reinterpret_cast<const float4*>(sv.dxr + blk(st, tiles, tile, net.h_t))[i];
    }
    __syncthreads();
    // the gradient that enters a state produced by the dynamics net is halved (pipeline.py:584); h_0 is not hooked
    normalize_bwd(lds + o.G, k >= 0 ? 0.5f : 1.0f, lds + o.HN, k >= 0 ? lds + o.R : nullptr, lds + o.HS, net.H, net.h_t, tid);
    __syncthreads();
    save_T_from_pk(lds + o.HS, net.h_t, (k >= 0 ? sv.dz_dyn1 : sv.dz_rep1) + blk(st, tiles, tile, net.h_t), tid);
    float* dzb = (k >= 0 ? sv.dz_dyn0 : sv.dz_rep0) + blk(st, tiles, tile, net.p_t);
    wide_mma<F, 4>(w1, net.L[l1].wtp, nullptr, net.p_t, net.h_t, lds + o.HS, wave, lane, [&](int t, f32x4 a) {
        f32x4 h;
        if (pre) {
            const int jj = (t - wave) / LW;
            h = jj == 0 ? gate[0] : (jj == 1 ? gate[1] : (jj == 2 ? gate[2] : gate[3]));
        } else {
            h = g_get_T(h1b, t, lane);
        }
        const f32x4 r = {h[0] > 0.0f ? a[0] : 0.0f, h[1] > 0.0f ? a[1] : 0.0f, h[2] > 0.0f ? a[2] : 0.0f, h[3] > 0.0f ? a[3] : 0.0f};
        g_put_T(dzb, t, lane, r);
        lds_put_T(lds + o.DZ, t, lane, r);
    }, NoPf{});
    if (k < 0) return;  // the observation needs no gradient
    __syncthreads();
    ks_mma<F>(w2, net.L[DYN0].wtp, nullptr, net.h_t, net.p_t, lds + o.DZ, lds + o.RED, wave, lane,
           [&](int t, f32x4 a) { lds_put_T(lds + o.G, t, lane, a); }, NoPf{});
    __syncthreads();
    copy_f4(sv.dxd + blk(k, tiles, tile, net.h_t) * DX_PARTS, lds + o.G, hf, tid);
}

// PERSISTENT CHAINS (register-resident form only).  The dynamics chain of a tile depends on nothing but the tile itself, and every one
// of its K stages multiplies the SAME two weight matrices (the dynamics net is shared by the unroll steps, pipeline.py:579-592).  A stage
// kernel per step re-streams 288 KB of operands per workgroup and pays a launch boundary and a prologue per step -- with one workgroup
// per CU nothing hides it: 13-15 us per stage at batch 4096 against 4 us of MFMA time.  Here a workgroup loads the operands ONCE and
// runs the whole chain of its tile(s) -- grid = min(tiles, CUs), tiles strided over the workgroups -- h_{k+1} going to the next stage
// through LDS.  Same arithmetic and summation orders as the stage kernels (bit-identical: tests/test_gpu_hip_learner.py).
//   k_learn_dyn_chain:       h_0 (from k_learn_repr) -> dynamics_0 .. dynamics_{K-1}: x_k, h1_k, u_{k+1}, h_{k+1}
//   k_learn_dyn_back_chain:  dL/dh_K .. dL/dh_1 (+ the heads' contributions) -> dynamics_{K-1} .. dynamics_0 backward -> dL/dh_0's
//                            dynamics part (the representation stage, k == -1, stays k_learn_back)
__global__ __launch_bounds__(LT) void k_learn_dyn_chain(LNet net, LSave sv, LBatch bt, LLds o) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), tiles = bt.tiles;
    const int K = net.K, hf = net.h_t * 64;
    const int e = tid >> 5, j = tid & 31;
    WideW w1;
    KsW w2;
    wide_load<6, true, true>(w1, net.L[DYN0].wp, net.L[DYN0].b, net.p_t, net.h_t + net.a_t, wave, lane);
    ks_load<true, true>(w2, net.L[DYN1].wp, net.L[DYN1].b, net.h_t, net.p_t, wave, lane);
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const float* hsrc = sv.hc + blk(0, tiles, tile, net.h_t);
        const float4 xv = chain_ld(hsrc, hf, tid);
        int act = sv.actc[tile * TILE + e];
        __syncthreads();  // (the previous tile's last stage is done with X)
        chain_st<true>(lds + o.X, hsrc, xv, hf, tid);
        for (int k = 0; k < K; k++) {
            // X <- [h_k | onehot(a_k)] (network.py:191-193); h_k is in place
            for (int a = j; a < net.a_t * 16; a += 32) lds[o.X + pk(net.h_t * 16 + a, e)] = a == act ? 1.0f : 0.0f;
            const int kn = k + 1 < K ? k + 1 : k;
            act = sv.actc[(kn * tiles + tile) * TILE + e];  // the next stage's, requested a stage ahead
            __syncthreads();
            save_T_from_pk(lds + o.X, net.h_t + net.a_t, sv.x + blk(k, tiles, tile, net.h_t + net.a_t), tid);
            float* h1b = sv.h1_dyn + blk(k, tiles, tile, net.p_t);
            wide_mma<true, 6>(w1, net.L[DYN0].wp, net.L[DYN0].b, net.p_t, net.h_t + net.a_t, lds + o.X, wave, lane, [&](int t, f32x4 a) {
                const f32x4 r = {fmaxf(a[0], 0.0f), fmaxf(a[1], 0.0f), fmaxf(a[2], 0.0f), fmaxf(a[3], 0.0f)};
                g_put_T(h1b, t, lane, r);
                lds_put_T(lds + o.H1, t, lane, r);
            }, NoPf{});
            __syncthreads();
            ks_mma<true>(w2, net.L[DYN1].wp, net.L[DYN1].b, net.h_t, net.p_t, lds + o.H1, lds + o.RED, wave, lane,
                         [&](int t, f32x4 a) { lds_put_T(lds + o.HN, t, lane, a); }, NoPf{});
            __syncthreads();
            normalize_fwd(lds + o.HN, lds + o.HS, net.H, net.h_t, tid);
            __syncthreads();
            const size_t cb = blk(k + 1, tiles, tile, net.h_t);
            copy_f4(sv.uc + cb, lds + o.HN, hf, tid);
            copy_f4(sv.hc + cb, lds + o.HS, hf, tid);
            if (tid < hf) reinterpret_cast<float4*>(lds + o.X)[tid] = reinterpret_cast<const float4*>(lds + o.HS)[tid];  // the next stage's input
        }
    }
}

__global__ __launch_bounds__(LT) void k_learn_dyn_back_chain(LNet net, LSave sv, LBatch bt, LLds o) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), tiles = bt.tiles;
    const int K = net.K, hf = net.h_t * 64;
    const int ci = tid < hf ? tid : hf - 1;
    const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    WideW w1;
    KsW w2;
    wide_load<4, false, true>(w1, net.L[DYN1].wtp, nullptr, net.p_t, net.h_t, wave, lane);
    ks_load<false, true>(w2, net.L[DYN0].wtp, nullptr, net.h_t, net.p_t, wave, lane);
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        // stage inputs, requested one stage ahead: the heads' contributions to dL/dh_{k+1} (policy, value: none for k + 1 == K), u_{k+1},
        // the reward head's dL/du_{k+1}, and the ReLU gate of this wave's tiles of dynamics_k's first layer
        float4 gb = z4, gc = z4, uu, rr;
        f32x4 gate[4];
        auto request = [&](int k) {
            const int jj = k + 1;
            const size_t cj = blk(jj, tiles, tile, net.h_t), cjc = jj < K ? cj : 0;
            gb = reinterpret_cast<const float4*>(sv.dxp + cjc)[ci];
            gc = reinterpret_cast<const float4*>(sv.dxv + cjc)[ci];
            uu = reinterpret_cast<const float4*>(sv.uc + cj)[ci];
            rr = reinterpret_cast<const float4*>(sv.dxr + blk(k, tiles, tile, net.h_t))[ci];
            const float* h1b = sv.h1_dyn + blk(k, tiles, tile, net.p_t);
#pragma unroll
            for (int i = 0; i < 4; i++) gate[i] = g_get_T(h1b, wave + i * LW < net.p_t ? wave + i * LW : net.p_t - 1, lane);
        };
        request(K - 1);
        __syncthreads();  // (the previous tile's last stage is done with G)
        if (tid < hf) reinterpret_cast<float4*>(lds + o.G)[tid] = z4;  // dL/dh_K: no consumer of h_K inside the loss
        for (int k = K - 1; k >= 0; k--) {
            const bool cons = k + 1 < K;
            if (tid < hf) {
                // G holds the dynamics part of dL/dh_{k+1} (the previous stage's result); add the heads' parts in the stage kernels' order
                float4* G4 = reinterpret_cast<float4*>(lds + o.G);
                const float4 ga = G4[tid];
                G4[tid] = cons ? make_float4((ga.x + gb.x) + gc.x, (ga.y + gb.y) + gc.y, (ga.z + gb.z) + gc.z, (ga.w + gb.w) + gc.w) : z4;
                reinterpret_cast<float4*>(lds + o.HN)[tid] = uu;
                reinterpret_cast<float4*>(lds + o.R)[tid] = rr;
            }
            f32x4 gk[4];
#pragma unroll
            for (int i = 0; i < 4; i++) gk[i] = gate[i];
            __syncthreads();
            if (k > 0) request(k - 1);
            normalize_bwd(lds + o.G, 0.5f, lds + o.HN, lds + o.R, lds + o.HS, net.H, net.h_t, tid);  // (halved: pipeline.py:584)
            __syncthreads();
            save_T_from_pk(lds + o.HS, net.h_t, sv.dz_dyn1 + blk(k, tiles, tile, net.h_t), tid);
            float* dzb = sv.dz_dyn0 + blk(k, tiles, tile, net.p_t);
            wide_mma<true, 4>(w1, net.L[DYN1].wtp, nullptr, net.p_t, net.h_t, lds + o.HS, wave, lane, [&](int t, f32x4 a) {
                const int jj = (t - wave) / LW;
                const f32x4 h = jj == 0 ? gk[0] : (jj == 1 ? gk[1] : (jj == 2 ? gk[2] : gk[3]));
                const f32x4 r = {h[0] > 0.0f ? a[0] : 0.0f, h[1] > 0.0f ? a[1] : 0.0f, h[2] > 0.0f ? a[2] : 0.0f, h[3] > 0.0f ? a[3] : 0.0f};
                g_put_T(dzb, t, lane, r);
                lds_put_T(lds + o.DZ, t, lane, r);
            }, NoPf{});
            __syncthreads();
            ks_mma<true>(w2, net.L[DYN0].wtp, nullptr, net.h_t, net.p_t, lds + o.DZ, lds + o.RED, wave, lane,
                         [&](int t, f32x4 a) { lds_put_T(lds + o.G, t, lane, a); }, NoPf{});
            __syncthreads();
        }
        // dL/dh_0's dynamics part for the representation stage (k_learn_back, k == -1)
        copy_f4(sv.dxd + blk(0, tiles, tile, net.h_t) * DX_PARTS, lds + o.G, hf, tid);
    }
}

// The same stage for SMALL batches, cut across the planes: workgroup (tile, part) owns the plane tiles [part p_t / np, (part + 1) p_t / np)
// -- one per wave -- of dL/dz1 and leaves a PARTIAL dL/dh_k over its slice; the next stage adds the np partials when it loads them (a
// kernel boundary is the cheapest grid-wide exchange on this part, and the consumer's sum costs np - 1 more 16-byte loads per lane).
// At batch 128 the backward chain is 6 dependent stages on 8 workgroups; sliced four ways each stage does a quarter of the MFMAs on
// 32 workgroups.  Register-resident forms only (the launcher checks: p_t / np <= LW, h_t <= 4, LW / h_t divides the slice).
template <bool F>
__global__ __launch_bounds__(LT) void k_learn_back_sliced(LNet net, LSave sv, LBatch bt, LLds o, int k) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), tile = blockIdx.x, tiles = bt.tiles;
    const int part = blockIdx.y, np = gridDim.y, K = net.K, j = k + 1, hf = net.h_t * 64;
    const int l0 = k >= 0 ? DYN0 : REP0, l1 = l0 + 1, st = k >= 0 ? k : 0;
    const int tb = part * net.p_t / np, te = (part + 1) * net.p_t / np;  // this workgroup's plane tiles
    const size_t cj = blk(j, tiles, tile, net.h_t);
    const int ci = tid < hf ? tid : hf - 1;
    const bool cons = j < K;
    const size_t cjc = cons ? cj : 0;
    float4 ga = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int pp = 0; pp < DX_PARTS; pp++) {  // (clamped, never guarded: see LOAD ORDER)
        const float4 v = reinterpret_cast<const float4*>(sv.dxd + (cjc * DX_PARTS) + (size_t)(pp < sv.dx_parts ? pp : 0) * net.h_t * 256)[ci];
        const float m = pp < sv.dx_parts ? 1.0f : 0.0f;
        ga = make_float4(ga.x + m * v.x, ga.y + m * v.y, ga.z + m * v.z, ga.w + m * v.w);
    }
    const float4 gb = reinterpret_cast<const float4*>(sv.dxp + cjc)[ci], gc = reinterpret_cast<const float4*>(sv.dxv + cjc)[ci];
    const float4 uu = reinterpret_cast<const float4*>(sv.uc + cj)[ci];
    const float4 rr = reinterpret_cast<const float4*>(sv.dxr + blk(st, tiles, tile, net.h_t))[ci];
    // operands: this wave's plane tile of W2^T (h_t blocks), its relu gate, and its share of W1^T over the slice
    const int t = tb + wave < te ? tb + wave : te - 1;
    const bool t_ok = tb + wave < te;
    const float4* W2T = reinterpret_cast<const float4*>(net.L[l1].wtp) + lane;
    float4 w1[4];
#pragma unroll
    for (int d = 0; d < 4; d++) w1[d] = ldg4(W2T + ((size_t)t * net.h_t + (d < net.h_t ? d : net.h_t - 1)) * 64);
    const float* h1b = (k >= 0 ? sv.h1_dyn : sv.h1_rep) + blk(st, tiles, tile, net.p_t);
    const f32x4 gate = g_get_T(h1b, t, lane);
    const int kparts = LW / net.h_t, ot = wave % net.h_t, kp = wave / net.h_t;  // output tile and reduction share of this wave in the second GEMM
    const int nb = (te - tb) / kparts, g0 = tb + kp * nb;
    const float4* W1T = reinterpret_cast<const float4*>(net.L[DYN0].wtp) + ((size_t)ot * net.p_t) * 64 + lane;
    float4 w2[4];
#pragma unroll
    for (int d = 0; d < 4; d++) w2[d] = ldg4(W1T + (size_t)(g0 + (d < nb ? d : nb - 1)) * 64);
    if (tid < hf) {
        reinterpret_cast<float4*>(lds + o.G)[tid] = cons ? make_float4((ga.x + gb.x) + gc.x, (ga.y + gb.y) + gc.y, (ga.z + gb.z) + gc.z, (ga.w + gb.w) + gc.w)
                                                         : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        reinterpret_cast<float4*>(lds + o.HN)[tid] = uu;
        reinterpret_cast<float4*>(lds + o.R)[tid] = rr;
    }
    __syncthreads();
    normalize_bwd(lds + o.G, k >= 0 ? 0.5f : 1.0f, lds + o.HN, k >= 0 ? lds + o.R : nullptr, lds + o.HS, net.H, net.h_t, tid);
    __syncthreads();
    if (part == 0) save_T_from_pk(lds + o.HS, net.h_t, (k >= 0 ? sv.dz_dyn1 : sv.dz_rep1) + blk(st, tiles, tile, net.h_t), tid);
    float* dzb = (k >= 0 ? sv.dz_dyn0 : sv.dz_rep0) + blk(st, tiles, tile, net.p_t);
    if (t_ok) {
        const float4* X4 = reinterpret_cast<const float4*>(lds + o.HS) + lane;
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int d = 0; d < 4; d++)
            if (d < net.h_t) a = mfma4(X4[d * 64], w1[d], a);
        const f32x4 r = {gate[0] > 0.0f ? a[0] : 0.0f, gate[1] > 0.0f ? a[1] : 0.0f, gate[2] > 0.0f ? a[2] : 0.0f, gate[3] > 0.0f ? a[3] : 0.0f};
        g_put_T(dzb, t, lane, r);
        lds_put_T(lds + o.DZ, t, lane, r);
    }
    if (k < 0) return;
    __syncthreads();
    {
        const float4* X4 = reinterpret_cast<const float4*>(lds + o.DZ) + lane;
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int d = 0; d < 4; d++)
            if (d < nb) a = mfma4(X4[(g0 + d) * 64], w2[d], a);
        reinterpret_cast<float4*>(lds + o.RED)[(kp * net.h_t + ot) * 64 + lane] = make_float4(a[0], a[1], a[2], a[3]);
    }
    __syncthreads();
    if (wave < net.h_t) {
        f32x4 sacc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int q = 0; q < kparts; q++) {
            const float4 v = reinterpret_cast<const float4*>(lds + o.RED)[(q * net.h_t + wave) * 64 + lane];
            sacc = sacc + f32x4{v.x, v.y, v.z, v.w};
        }
        lds_put_T(lds + o.G, wave, lane, sacc);
    }
    __syncthreads();
    copy_f4(sv.dxd + blk(k, tiles, tile, net.h_t) * DX_PARTS + (size_t)part * net.h_t * 256, lds + o.G, hf, tid);
}

// loss = mean_i w_i sum_k (reward + value + policy loss) (pipeline.py:594-597) from the heads' per-workgroup partial sums; fixed summation
// order: LT strided partial sums (four independent chains per thread: the loads of a chain step are in flight together), then a tree.
// Runs as ONE EXTRA WORKGROUP of the weight-gradient launch (the last block index): as a kernel of its own it cost a launch and a
// dependent boundary for 5 us of work.  `sc`: LT floats of LDS.
__device__ __forceinline__ void loss_finish(const float* __restrict__ lossp, int n, int B, float* __restrict__ loss_out, float* sc) {
    const int tid = threadIdx.x;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int i = tid;
    for (; i + 3 * LT < n; i += 4 * LT) {
        const float a = lossp[i], b = lossp[i + LT], c = lossp[i + 2 * LT], d = lossp[i + 3 * LT];
        s0 += a; s1 += b; s2 += c; s3 += d;
    }
    for (; i < n; i += LT) s0 += lossp[i];
    sc[tid] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int m = LT / 2; m > 0; m >>= 1) {
        if (tid < m) sc[tid] += sc[tid + m];
        __syncthreads();
    }
    if (tid == 0) loss_out[0] = sc[0] / (float)B;
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradients
// ------------------------------------------------------------------------------------------------------------------
struct DwJob {
    const float* a;  // dL/dz blocks (rows of W)
    const float* b;  // layer-input blocks (columns of W)
    int gw, gb;      // float offsets of the weight / bias gradient in the flat gradient vector (state_dict order); gb < 0: no bias
    int a_ft, b_ft;  // feature tiles per block of a / b
    int a_t0, na, b_t0, nb;  // tile ranges; na * nb <= 4
    int R;           // reduction blocks = steps * tiles
    int n, k;        // real rows / columns
    int kH, kHpad;   // column c of a block is W column (c < kH ? c : c - kHpad + kH); c in [kH, kHpad) is padding
};

// dW[n][c] = sum_r sum_s dz_r[n][s] x_r[c][s]: A operand = a T block of dz (16 rows x 4 samples per k-step), B operand = a T block of x.
// The workgroup's waves split the reduction blocks; partial tiles meet in LDS and are added in wave order (deterministic).
// grid.y = reduction split: slice y accumulates blocks [y R / ny, (y + 1) R / ny) into gradient slice y.
__global__ __launch_bounds__(LT) void k_learn_dw(const DwJob* __restrict__ jobs, float* __restrict__ grads, size_t grad_stride, const float* __restrict__ lossp,
                                                 int n_loss, int B, float* __restrict__ loss_out) {
    extern __shared__ __align__(16) float red[];  // LW x 8 accumulator tiles
    if (blockIdx.x == gridDim.x - 1) {  // the extra workgroup: the loss (loss_finish)
        if (blockIdx.y == 0) loss_finish(lossp, n_loss, B, loss_out, red);
        return;
    }
    const DwJob J = jobs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = (int)((long long)blockIdx.y * J.R / gridDim.y), r1 = (int)((long long)(blockIdx.y + 1) * J.R / gridDim.y);
    float* G = grads + (size_t)blockIdx.y * grad_stride;
    const bool bias = J.gb >= 0;
    f32x4 acc[4], accb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = accb[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const float one = (lane & 15) == 0 ? 1.0f : 0.0f;
    const float4* A4 = reinterpret_cast<const float4*>(J.a) + lane;
    const float4* B4 = reinterpret_cast<const float4*>(J.b) + lane;
    if (J.na == 1) {
        for (int r = r0 + wave; r < r1; r += LW) {
            const float4 a = ldg4(A4 + ((size_t)r * J.a_ft + J.a_t0) * 64);
            float4 b[4];
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = ldg4(B4 + ((size_t)r * J.b_ft + J.b_t0 + (j < J.nb ? j : 0)) * 64);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (j < J.nb) acc[j] = mfma4(a, b[j], acc[j]);
            if (bias) accb[0] = mfma4(a, make_float4(one, one, one, one), accb[0]);
        }
    } else {
        for (int r = r0 + wave; r < r1; r += LW) {
            const float4 b = ldg4(B4 + ((size_t)r * J.b_ft + J.b_t0) * 64);
            float4 a[4];
#pragma unroll
            for (int j = 0; j < 4; j++) a[j] = ldg4(A4 + ((size_t)r * J.a_ft + J.a_t0 + (j < J.na ? j : 0)) * 64);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (j < J.na) {
                    acc[j] = mfma4(a[j], b, acc[j]);
                    if (bias) accb[j] = mfma4(a[j], make_float4(one, one, one, one), accb[j]);
                }
        }
    }
    const int nacc = J.na * J.nb, nbias = bias ? J.na : 0;
    float4* R4 = reinterpret_cast<float4*>(red);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (j < nacc) R4[(wave * 8 + j) * 64 + lane] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
        if (j < nbias) R4[(wave * 8 + 4 + j) * 64 + lane] = make_float4(accb[j][0], accb[j][1], accb[j][2], accb[j][3]);
    }
    __syncthreads();
    // wave j adds accumulator j of all waves (j < 4: weight tiles, j >= 4: bias tiles) and writes it
    const int j = wave;
    if ((j < 4 && j < nacc) || (j >= 4 && j - 4 < nbias)) {
        f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int w = 0; w < LW; w++) {
            const float4 v = R4[(w * 8 + j) * 64 + lane];
            s = s + f32x4{v.x, v.y, v.z, v.w};
        }
        const int q = lane >> 4, c = lane & 15;
        if (j < 4) {
            const int ja = J.na == 1 ? 0 : j, jb = J.na == 1 ? j : 0;
            const int pc = 16 * (J.b_t0 + jb) + c;
            const int col = pc < J.kH ? pc : (pc >= J.kHpad ? pc - J.kHpad + J.kH : -1);
            if (col >= 0 && col < J.k) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int row = 16 * (J.a_t0 + ja) + 4 * q + i;
                    if (row < J.n) G[J.gw + (size_t)row * J.k + col] = s[i];
                }
            }
        } else if (c == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = 16 * (J.a_t0 + (j - 4)) + 4 * q + i;
                if (row < J.n) G[J.gb + row] = s[i];
            }
        }
    }
}

// The same gradients for LONG reductions (large batches).  There the small-block kernel above is bound by the CU's 64 B / clk of L1
// bandwidth, not by the MFMA pipes: a wave that owns 1 x 4 tiles loads 5 operand blocks (5 KiB) per 16 MFMAs.  Here a wave owns up to
// 4 x 5 tiles -- 9 KiB per 80 MFMAs -- runs its whole reduction slice alone (no LDS exchange; the slices meet in k_learn_gradsum), and
// requests the next reduction block's operands before multiplying the current one.  One unit (tile group, slice) per WAVE, and the
// eight waves of a workgroup hold the units of ONE layer and ONE slice: together they read every dz and every input block of that slice
// exactly once from HBM (the operand the units share -- the 4 input tiles of a 512 x 64 layer, the 4 dz tiles of a 64 x 512 layer --
// comes from L2 for seven of them).  With one unit per wave in layer-mixed workgroups (round 4's first form) the 245 MB of saved
// activations of a 4096-sample batch crossed the fabric 2.7 times, 3.9 TB/s for the kernel's 166 us: it was memory-bound at 40 % MFMA
// busy.  The bias gradient (row sums of dz) is accumulated on the vector ALUs under the MFMAs (it was a fifth MFMA column: +25 %).
struct DwBig {
    const float* a;
    const float* b;
    int gw, gb;
    int a_ft, b_ft;
    int a_t0, na, b_t0, nb;  // up to 4 x 5 tiles; na == 0: padding unit (the workgroup's layer has fewer than DWB_WAVES units)
    int R, n, k, kH, kHpad;
    int slice, nslices;      // this unit's share of the reduction: blocks [slice R / nslices, (slice + 1) R / nslices), written to gradient
                             // slice `slice` (one unit per job and slice)
};
constexpr int DWB_WAVES = 8, DWB_NB = 5;
// (Measured and not kept, round 4: the reduction loop instantiated per tile-group shape (NA x NB compile-time, no clamped loads, no guards)
// with the block's MFMAs issued component by component across the accumulators instead of four dependent k-steps per accumulator --
// 130 -> 158 us at batch 4096 whichever way the MFMAs were ordered: twenty loop bodies in one kernel and the register moves of their
// operand rotation cost more than the clamped loads and the 40-cycle dependent issue they remove.)
__global__ __launch_bounds__(DWB_WAVES * 64) void k_learn_dw_big(const DwBig* __restrict__ jobs, int njobs, float* __restrict__ grads, size_t grad_stride,
                                                                 const float* __restrict__ lossp, int n_loss, int B, float* __restrict__ loss_out) {
    static_assert(DWB_WAVES * 64 == LT, "loss_finish runs on LT threads");
    if (blockIdx.x == gridDim.x - 1) {  // the extra workgroup: the loss (loss_finish)
        __shared__ float sc[LT];
        loss_finish(lossp, n_loss, B, loss_out, sc);
        return;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int ji = blockIdx.x * DWB_WAVES + wave;
    if (ji >= njobs) return;
    const DwBig J = jobs[ji];
    if (J.na == 0) return;
    const int r0 = (int)((long long)J.slice * J.R / J.nslices), r1 = (int)((long long)(J.slice + 1) * J.R / J.nslices);
    float* G = grads + (size_t)J.slice * grad_stride;
    const bool bias = J.gb >= 0;
    f32x4 acc[4][DWB_NB];
    float accb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        accb[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < DWB_NB; j++) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    const float4* A4 = reinterpret_cast<const float4*>(J.a) + lane;
    const float4* B4 = reinterpret_cast<const float4*>(J.b) + lane;
    // tile offsets inside a block, clamped to the job's real tiles (a clamped tile is loaded, never multiplied)
    int ao[4], bo[DWB_NB];
#pragma unroll
    for (int i = 0; i < 4; i++) ao[i] = (J.a_t0 + (i < J.na ? i : J.na - 1)) * 64;
#pragma unroll
    for (int i = 0; i < DWB_NB; i++) bo[i] = (J.b_t0 + (i < J.nb ? i : J.nb - 1)) * 64;
    // three operand sets in rotation: block r is multiplied while r + 1 is landing and r + 2 is requested -- the kernel streams the saved
    // activations of the whole batch from HBM (0.4 GB at 4096 samples) with 16 waves per CU, and one block ahead left too few bytes in
    // flight to keep the memory side busy
    float4 a[3][4], b[3][DWB_NB];
    auto request = [&](int set, int rr) {
        const int rc = rr < r1 ? rr : r1 - 1;
#pragma unroll
        for (int i = 0; i < 4; i++) a[set][i] = ldg4g(A4 + (size_t)rc * J.a_ft * 64 + ao[i]);
#pragma unroll
        for (int i = 0; i < DWB_NB; i++) b[set][i] = ldg4g(B4 + (size_t)rc * J.b_ft * 64 + bo[i]);
    };
    auto multiply = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (i < J.na) {
#pragma unroll
                for (int j = 0; j < DWB_NB; j++)
                    if (j < J.nb) acc[i][j] = mfma4(a[set][i], b[set][j], acc[i][j]);
                if (bias) accb[i] = accb[i] + (((a[set][i].x + a[set][i].y) + a[set][i].z) + a[set][i].w);  // this lane's 4 samples of row (lane & 15)
            }
        }
    };
    if (r0 < r1) {
        request(0, r0);
        request(1, r0 + 1);
    }
    for (int r = r0; r < r1; r += 3) {
        request(2, r + 2);
        multiply(0);
        if (r + 1 < r1) {
            request(0, r + 3);
            multiply(1);
        }
        if (r + 2 < r1) {
            request(1, r + 4);
            multiply(2);
        }
    }
    const int q = lane >> 4, c = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i >= J.na) continue;
#pragma unroll
        for (int j = 0; j < DWB_NB; j++) {
            if (j >= J.nb) continue;
            const int pc = 16 * (J.b_t0 + j) + c;
            const int col = pc < J.kH ? pc : (pc >= J.kHpad ? pc - J.kHpad + J.kH : -1);
            if (col < 0 || col >= J.k) continue;
#pragma unroll
            for (int ii = 0; ii < 4; ii++) {
                const int row = 16 * (J.a_t0 + i) + 4 * q + ii;
                if (row < J.n) G[J.gw + (size_t)row * J.k + col] = acc[i][j][ii];
            }
        }
        if (bias) {  // the four sample groups of a row sit 16 lanes apart: (g0 + g1) + (g2 + g3)
            float t = accb[i];
            t = t + __shfl_xor(t, 16, 64);
            t = t + __shfl_xor(t, 32, 64);
            const int row = 16 * (J.a_t0 + i) + c;
            if (q == 0 && row < J.n) G[J.gb + row] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// loss, gradient norm, optimizer
// ------------------------------------------------------------------------------------------------------------------
struct LTensor { int off, n, k, layer, is_bias; };
struct LParams {
    LTensor t[2 * NLAYER];
    float* wp[NLAYER];
    float* wtp[NLAYER];
    float* b[NLAYER];
    int nt[NLAYER], kg[NLAYER];
    int kH[NLAYER], kHpad[NLAYER];  // column map of the layer's input (dynamics layer 0: [hidden | padding | actions])
    int total;
};

// sum of the gradient slices -> slice 0 (when the weight-gradient kernel ran with a reduction split), and per-block partial sums of
// squares for clip_grad_norm_ (pipeline.py:246-247)
__global__ __launch_bounds__(256) void k_learn_gradsum(float* __restrict__ grads, size_t stride, int slices, int total, float* __restrict__ sq_part) {
    __shared__ float sc[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float g = 0.0f;
    if (i < total) {
        g = grads[i];
        for (int s = 1; s < slices; s++) g += grads[(size_t)s * stride + i];
        if (slices > 1) grads[i] = g;
    }
    float v = g * g;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) sq_part[blockIdx.x] = (sc[0] + sc[1]) + (sc[2] + sc[3]);
}

struct AdamArgs {
    float lr, beta1, beta2, eps, weight_decay, max_norm;  // max_norm <= 0: no clipping
    float bc1, bc2;                                       // 1 - beta^t
    int sq_blocks;                                        // partial sums of squares written by k_learn_gradsum
    int pack_only;                                        // 1: only (re)build the operand copies from the master weights
};

// torch.optim.Adam (weight_decay added to the gradient, as classic/run_training.py:94 builds it) on the master weights in
// state_dict layout, then the element's slots in the MFMA operand copies.  grid (blocks, tensor).
__global__ __launch_bounds__(256) void k_learn_adam(LParams P, float* __restrict__ params, const float* __restrict__ grads, float* __restrict__ m,
                                                     float* __restrict__ v, const float* __restrict__ sq_part, AdamArgs a) {
    const LTensor T = P.t[blockIdx.y];
    const int cnt = T.is_bias ? T.n : T.n * T.k;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= cnt) return;
    float coef = 1.0f;
    if (!a.pack_only && a.max_norm > 0.0f) {  // every block adds the same partials in the same order
        float s = 0.0f;
        for (int b = 0; b < a.sq_blocks; b++) s += sq_part[b];
        const float c = a.max_norm / (sqrtf(s) + 1e-6f);
        coef = c < 1.0f ? c : 1.0f;
    }
    if (i >= cnt) return;
    const int gi = T.off + i;
    float p = params[gi];
    if (!a.pack_only) {
        float g = grads[gi] * coef;
        g = g + a.weight_decay * p;
        const float mi = a.beta1 * m[gi] + (1.0f - a.beta1) * g;
        const float vi = a.beta2 * v[gi] + (1.0f - a.beta2) * g * g;
        m[gi] = mi;
        v[gi] = vi;
        const float denom = sqrtf(vi) / sqrtf(a.bc2) + a.eps;
        p = p - (a.lr / a.bc1) * (mi / denom);
        params[gi] = p;
    }
    const int l = T.layer;
    if (T.is_bias) {
        P.b[l][i] = p;
        return;
    }
    const int nn = i / T.k, kk = i - nn * T.k;
    const int pc = kk < P.kH[l] ? kk : kk - P.kH[l] + P.kHpad[l];
    P.wp[l][(((size_t)(nn >> 4) * P.kg[l] + (pc >> 4)) * 64 + ((pc >> 2) & 3) * 16 + (nn & 15)) * 4 + (pc & 3)] = p;
    P.wtp[l][(((size_t)(pc >> 4) * P.nt[l] + (nn >> 4)) * 64 + ((nn >> 2) & 3) * 16 + (pc & 15)) * 4 + (nn & 3)] = p;
}

}  // namespace mzl
