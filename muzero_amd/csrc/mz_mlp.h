// mz_mlp.h -- fused MuZero MLP inference for one workgroup tile of 16 environments (gfx950, wave64).
//
// Every Linear layer (network.py:145-149,172-182,212-222) is Y[n][e] = b[n] + sum_k W[n][k] * X[e][k] computed with
// v_mfma_f32_16x16x4_f32: A operand = 16 output neurons x 4 k of the weight matrix, B operand = 4 k x 16 environments
// of the activations, D = 16 neurons x 16 environments.  The MFMA accumulates its 4 k in order as one float32 fmaf
// chain (exact f32, no wider internal sum).
//
// SUMMATION ORDER (the numerical contract; the CPU checker restates the same order):
//   * inputs are taken in blocks of 16; k-step i (0..3) of block g multiplies k = 16g + 4q + i, q = 0..3 in that order.
//     That is the order in which an MFMA consumes a D-layout accumulator as its B operand: lane (e, q) of D holds
//     neurons 16t + 4q .. 4q+3 in registers 0..3, and register i of lane (e, q) is B[k = q][n = e] of k-step i.  A
//     layer's output therefore feeds the next layer straight from registers (mz_search_fast.h) -- or from LDS, where
//     the same order makes lane (e, q)'s four k-steps ONE aligned float4 of four consecutive k:
//         pk(k, e) = ((k >> 4) * 64 + ((k >> 2) & 3) * 16 + e) * 4 + (k & 3)          [float index]
//   * the one-hot action inputs of the dynamics net (network.py:191-193) form their own block(s) after the (padded)
//     hidden blocks, in natural order: action a of a block sits at lane q = a & 3, k-step i = a >> 2 (pk_act).
//   * the second layer of every two-layer net (K = num_planes) is K-SPLIT: its blocks are dealt to 4 contiguous
//     quarters, one chain each (quarter 0 starts from the bias, the others from +0), result ((c0 + c1) + c2) + c3.
//     In the tuned kernel quarter w is the part of the hidden layer that wave w computed and still holds in registers.
// Weights are pre-packed on the host into the matching A-operand order, one global_load_dwordx4 per lane (1 KiB
// contiguous per wave) per tile and block:
//     Wp[((t * KG + g) * 64 + lane) * 4 + i] = W[16 t + (lane & 15)][kidx(g, q = lane >> 4, i)]
// Weights are streamed L2 -> VGPR (each element is used once per 16-env tile, so LDS staging would add nothing).
#pragma once
#include "mz_device.h"

namespace mz {

#ifdef MZ_STAMPS
// diagnostic build only: cycle sums of the root inference's segments (thread 0 of block 0), read by tools/phase_profile.py
__device__ long long g_root_ts[8];
#define MZ_ROOT_TS_START() long long _rt0 = (blockIdx.x == 0 && threadIdx.x == 0) ? (long long)__builtin_readcyclecounter() : 0
#define MZ_ROOT_TS(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long _n = __builtin_readcyclecounter(); g_root_ts[i] += _n - _rt0; _rt0 = _n; } } while (0)
#else
#define MZ_ROOT_TS_START() do {} while (0)
#define MZ_ROOT_TS(i) do {} while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE_E = 16;      // environments per workgroup tile
constexpr int WG_THREADS = 256; // 4 waves
constexpr int WG_WAVES = 4;

__device__ __forceinline__ int pk(int k, int e) { return (((k >> 4) * 64 + ((k >> 2) & 3) * 16 + e) << 2) + (k & 3); }
// one-hot action a of the dynamics input; hblocks = 16-blocks of the padded hidden part in front of the action block(s)
__device__ __forceinline__ int pk_act(int a, int e, int hblocks) {
    return (((hblocks + (a >> 4)) * 64 + (a & 3) * 16 + e) << 2) + ((a & 15) >> 2);
}

// Layer ids in the packed parameter table (state_dict order, network.py:236-267)
enum { L_REP0 = 0, L_REP1, L_DYN0, L_DYN1, L_REW0, L_REW1, L_POL0, L_POL1, L_VAL0, L_VAL1, L_COUNT };

struct MlpLayer {
    const float* w;  // packed [n_tiles][kg][64][4]
    const float* b;  // [n_tiles*16], zero padded
    int n;           // real output features
    int k;           // real input features
    int n_tiles;     // ceil(n/16)
    int kg;          // 16-blocks of the (padded) input; the dynamics layer 0 counts its action block(s) too
    int last_steps;  // k-steps of the LAST block (1..4); every other block runs all 4 (padding is exact zeros)
    int split;       // K-split layer: 4 quarter chains of kq blocks each
    int kq;          // ceil(kg / 4)
    int b_lds;       // float offset of the LDS copy of b (staged once per kernel by stage_biases)
};

struct MlpNet {
    MlpLayer L[L_COUNT];
    int in_dim, A, P, H, Sv, Sr;
    int in_pad, x_pad, p_pad, h_pad;  // multiples of 16: obs, pad16(H) + pad16(A), P, H
    const float* b_all;  // every layer's padded bias vector, concatenated in the order of the LDS copies (b_lds)
    int b_base, b_count; // first float of the LDS copy, number of floats
};

// LDS carve-out of the network part (float offsets from the dynamic-LDS base)
struct MlpLds {
    int X;    // [x_pad or in_pad][16] packed : layer input (obs or hidden+onehot)
    int H1;   // [p_pad][16] packed            : first hidden layer (also reward/policy head hidden)
    int V1;   // [p_pad][16] packed            : value head hidden
    int HN;   // [h_pad][16] packed            : un-normalised hidden state
    int HS;   // [h_pad][16] packed            : normalised hidden state
    int LG;   // [2][16][lg_stride]            : head logits (0: reward / policy, 1: value)
    int lg_stride;
    int OUT;  // [16][4] : per-env scalars {reward, value, -, -}
    int BIAS; // all layers' padded biases (MlpLayer::b_lds)
    int PM;   // [2][4][16] : per-wave min / max partials of the hidden state (fused normalisation)
    int total_floats;
};

// one global_load_dwordx4: read through HIP's float4 struct the four components are separate scalar loads to the optimiser,
// which re-merges them only some of the time -- the root inference's K-split loops ran on dword loads with one 64-bit
// address each (2.5x slower than their MFMAs)
__device__ __forceinline__ float4 ldg4(const float4* p) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    return make_float4(v[0], v[1], v[2], v[3]);
}

// 4 k-steps (or `steps` of them) of one block for NACC accumulators
template <int NACC>
__device__ __forceinline__ void mma_block(f32x4 (&acc)[NACC], const float4 (&w)[NACC], const float4 x, int steps) {
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].x, x.x, acc[j], 0, 0, 0);
    if (steps > 1) {
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].y, x.y, acc[j], 0, 0, 0);
    }
    if (steps > 2) {
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].z, x.z, acc[j], 0, 0, 0);
    }
    if (steps > 3) {
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].w, x.w, acc[j], 0, 0, 0);
    }
}

// all 4 k-steps of a block: no step-count branches between the MFMAs (every block but a layer's last one is full)
template <int NACC>
__device__ __forceinline__ void mma_block_full(f32x4 (&acc)[NACC], const float4 (&w)[NACC], const float4 x) {
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].x, x.x, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].y, x.y, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].z, x.z, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].w, x.w, acc[j], 0, 0, 0);
}

// One Linear layer over this workgroup's tile.  Tiles of 16 output neurons are dealt to waves round-robin
// (tile = wave_slot + i*WG_WAVES); up to NACC tiles are accumulated concurrently (independent MFMA chains hide the
// 40-cycle dependent latency of v_mfma_f32_16x16x4_f32).  epi(tile, acc) receives D: acc[r] = Y[16*tile + 4q + r][e].
template <int NACC, typename Epi>
__device__ __forceinline__ void gemm_chunk(const MlpLayer& L, const float* __restrict__ lds, const float* __restrict__ Xs, int t0, int lane,
                                           Epi& epi) {
    const int q = lane >> 4;
    f32x4 acc[NACC];
    const float4* wp[NACC];
#pragma unroll
    for (int j = 0; j < NACC; j++) {
        const int t = t0 + j * WG_WAVES;
        const float4 bv = *reinterpret_cast<const float4*>(lds + L.b_lds + t * 16 + q * 4);
        acc[j] = f32x4{bv.x, bv.y, bv.z, bv.w};
        wp[j] = reinterpret_cast<const float4*>(L.w) + (size_t)t * L.kg * 64 + lane;
    }
    const float4* xp = reinterpret_cast<const float4*>(Xs) + lane;
    // weights one block ahead of the MFMAs (an L2 round trip is ~2 k cycles; un-prefetched, every block would wait for it)
    // two blocks of weights in flight, in two statically named register sets (a ring rotated by register moves would
    // make every iteration wait for the load it has just issued); every block but the layer's last one is full
    float4 w0[NACC], w1[NACC];
    const int g1 = L.kg > 1 ? 1 : 0;
#pragma unroll
    for (int j = 0; j < NACC; j++) w0[j] = ldg4(&wp[j][0]);
#pragma unroll
    for (int j = 0; j < NACC; j++) w1[j] = ldg4(&wp[j][g1 * 64]);
    int g = 0;
    for (; g + 2 < L.kg; g += 2) {
        const float4 x0 = xp[g * 64], x1 = xp[(g + 1) * 64];
        const int gn = g + 3 < L.kg ? g + 3 : L.kg - 1;
        mma_block_full<NACC>(acc, w0, x0);
#pragma unroll
        for (int j = 0; j < NACC; j++) w0[j] = ldg4(&wp[j][(g + 2) * 64]);
        mma_block_full<NACC>(acc, w1, x1);
#pragma unroll
        for (int j = 0; j < NACC; j++) w1[j] = ldg4(&wp[j][gn * 64]);
    }
    if (L.kg - g == 2) {
        mma_block_full<NACC>(acc, w0, xp[g * 64]);
        mma_block<NACC>(acc, w1, xp[(g + 1) * 64], L.last_steps);
    } else {
        mma_block<NACC>(acc, w0, xp[g * 64], L.last_steps);
    }
#pragma unroll
    for (int j = 0; j < NACC; j++) epi(t0 + j * WG_WAVES, acc[j]);
}

// K-split layer: NT2 tiles x 4 quarter chains concurrently; quarter c covers blocks [c*kq, min(kg, (c+1)*kq))
template <int NT2, typename Epi>
__device__ __forceinline__ void gemm_chunk_split(const MlpLayer& L, const float* __restrict__ lds, const float* __restrict__ Xs, int t0,
                                                 int lane, Epi& epi) {
    const int q = lane >> 4;
    f32x4 acc[NT2][4];
    const float4* wp[NT2];
#pragma unroll
    for (int j = 0; j < NT2; j++) {
        const int t = t0 + j * WG_WAVES;
        const float4 bv = *reinterpret_cast<const float4*>(lds + L.b_lds + t * 16 + q * 4);
        acc[j][0] = f32x4{bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int c = 1; c < 4; c++) acc[j][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        wp[j] = reinterpret_cast<const float4*>(L.w) + (size_t)t * L.kg * 64 + lane;
    }
    const float4* xp = reinterpret_cast<const float4*>(Xs) + lane;
    constexpr int PD = NT2 == 1 ? 4 : 2;
#ifdef MZ_STAMPS
    const long long _gt0 = (blockIdx.x == 0 && threadIdx.x == 0) ? (long long)__builtin_readcyclecounter() : 0;
#endif
    if (L.kg == 4 * L.kq && L.last_steps == 4 && L.kq % PD == 0) {
        // K a multiple of 64 * PD (every shipped configuration): four equal quarters of full blocks; PD rounds of weights in
        // flight in statically named register sets (see gemm_chunk), straight-line body
        float4 wr[PD][4][NT2];
#pragma unroll
        for (int d = 0; d < PD; d++) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
#pragma unroll
                for (int j = 0; j < NT2; j++) wr[d][c][j] = ldg4(&wp[j][(c * L.kq + d) * 64]);
            }
        }
        for (int gg0 = 0; gg0 < L.kq; gg0 += PD) {
#pragma unroll
            for (int d = 0; d < PD; d++) {
                const int gg = gg0 + d, gw = gg + PD < L.kq ? gg + PD : L.kq - 1;  // the tail re-loads the last round
                float4 x[4];
#pragma unroll
                for (int c = 0; c < 4; c++) x[c] = xp[(c * L.kq + gg) * 64];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    f32x4 a[NT2];
#pragma unroll
                    for (int j = 0; j < NT2; j++) a[j] = acc[j][c];
                    mma_block_full<NT2>(a, wr[d][c], x[c]);
#pragma unroll
                    for (int j = 0; j < NT2; j++) acc[j][c] = a[j];
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
#pragma unroll
                    for (int j = 0; j < NT2; j++) wr[d][c][j] = ldg4(&wp[j][(c * L.kq + gw) * 64]);
                }
            }
        }
    } else {
        // any other shape: the four quarters' blocks of round gg + 1 are loaded while round gg multiplies
        float4 w[4][NT2], wn[4][NT2];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int g = c * L.kq < L.kg ? c * L.kq : 0;
#pragma unroll
            for (int j = 0; j < NT2; j++) w[c][j] = ldg4(&wp[j][g * 64]);
        }
        for (int gg = 0; gg < L.kq; gg++) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int g = c * L.kq + gg + 1;
                const int gl = (gg + 1 < L.kq && g < L.kg) ? g : 0;
#pragma unroll
                for (int j = 0; j < NT2; j++) wn[c][j] = ldg4(&wp[j][gl * 64]);
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int g = c * L.kq + gg;
                if (g < L.kg) {
                    const float4 x = xp[g * 64];
                    f32x4 a[NT2];
#pragma unroll
                    for (int j = 0; j < NT2; j++) a[j] = acc[j][c];
                    mma_block<NT2>(a, w[c], x, g + 1 < L.kg ? 4 : L.last_steps);
#pragma unroll
                    for (int j = 0; j < NT2; j++) acc[j][c] = a[j];
                }
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
#pragma unroll
                for (int j = 0; j < NT2; j++) w[c][j] = wn[c][j];
            }
        }
    }
#ifdef MZ_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) g_root_ts[7] += (long long)__builtin_readcyclecounter() - _gt0;  // K-split GEMM loops of wave 0 (diagnostic)
#endif
#pragma unroll
    for (int j = 0; j < NT2; j++) {
        const f32x4 y = ((acc[j][0] + acc[j][1]) + acc[j][2]) + acc[j][3];
        epi(t0 + j * WG_WAVES, y);
    }
}

// K-split layer with ONE output tile (a policy head with A <= 16): its four quarter chains dealt to the four waves -- on one
// wave, as gemm_chunk_split runs them, the other three idle for the whole layer.  The partial tiles meet in `scratch`
// (4 x 64 float4); wave 0 adds them in gemm_chunk_split's order, so the sums are the same.  Every thread of the workgroup
// must call this (one barrier inside); returns false, having done nothing, when the layer is not of that shape.
template <typename Epi>
__device__ __forceinline__ bool gemm_one_tile_across_waves(const MlpLayer& L, const float* __restrict__ lds, const float* __restrict__ Xs,
                                                           float* scratch, int wave, int lane, bool active, Epi epi) {
    constexpr int KQ = 8;  // blocks per quarter kept in flight at once
    if (!(L.split && L.n_tiles == 1 && L.kg == 4 * L.kq && L.last_steps == 4 && L.kq <= KQ)) return false;
    if (active) {
        const int q = lane >> 4;
        f32x4 acc[1];
        const float4 bv = *reinterpret_cast<const float4*>(lds + L.b_lds + q * 4);
        acc[0] = wave == 0 ? f32x4{bv.x, bv.y, bv.z, bv.w} : f32x4{0.0f, 0.0f, 0.0f, 0.0f};  // quarter 0 starts from the bias
        const float4* wp = reinterpret_cast<const float4*>(L.w) + (size_t)(wave * L.kq) * 64 + lane;
        const float4* xp = reinterpret_cast<const float4*>(Xs) + (size_t)(wave * L.kq) * 64 + lane;
        float4 w[KQ][1];
#pragma unroll
        for (int g = 0; g < KQ; g++) w[g][0] = ldg4(&wp[(g < L.kq ? g : 0) * 64]);
#pragma unroll
        for (int g = 0; g < KQ; g++)
            if (g < L.kq) mma_block_full<1>(acc, w[g], xp[g * 64]);
        reinterpret_cast<float4*>(scratch)[wave * 64 + lane] = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
    }
    __syncthreads();
    if (active && wave == 0) {
        const float4* sc = reinterpret_cast<const float4*>(scratch) + lane;
        const float4 c0 = sc[0], c1 = sc[64], c2 = sc[128], c3 = sc[192];
        const f32x4 y = {((c0.x + c1.x) + c2.x) + c3.x, ((c0.y + c1.y) + c2.y) + c3.y, ((c0.z + c1.z) + c2.z) + c3.z, ((c0.w + c1.w) + c2.w) + c3.w};
        epi(0, y);
    }
    return true;
}

template <typename Epi>
__device__ __forceinline__ void gemm_layer(const MlpLayer& L, const float* __restrict__ lds, const float* __restrict__ Xs, int wave_slot,
                                           int lane, Epi epi) {
    int t = wave_slot;
    if (L.split) {
        while (t + 1 * WG_WAVES < L.n_tiles) {
            gemm_chunk_split<2>(L, lds, Xs, t, lane, epi);
            t += 2 * WG_WAVES;
        }
        if (t < L.n_tiles) gemm_chunk_split<1>(L, lds, Xs, t, lane, epi);
        return;
    }
    while (t + 7 * WG_WAVES < L.n_tiles) {
        gemm_chunk<8>(L, lds, Xs, t, lane, epi);
        t += 8 * WG_WAVES;
    }
    if (t + 3 * WG_WAVES < L.n_tiles) {
        gemm_chunk<4>(L, lds, Xs, t, lane, epi);
        t += 4 * WG_WAVES;
    }
    if (t + 1 * WG_WAVES < L.n_tiles) {
        gemm_chunk<2>(L, lds, Xs, t, lane, epi);
        t += 2 * WG_WAVES;
    }
    if (t < L.n_tiles) gemm_chunk<1>(L, lds, Xs, t, lane, epi);
}

// epilogue: ReLU, store to the packed LDS buffer that feeds the next layer (neuron n becomes k = n there)
struct EpiReluPacked {
    float* dst;
    int lane;
    __device__ __forceinline__ void operator()(int t, const f32x4& a) const {
        // pk(16t + 4q + r, e), r = 0..3, is the float4 #(t*64 + lane): one ds_write_b128
        // ReLU as one v_med3_f32 (median of x, 0, +inf == max(x, 0) for every non-NaN x)
        const float inf = __uint_as_float(0x7f800000u);
        reinterpret_cast<float4*>(dst)[t * 64 + lane] =
            make_float4(__builtin_amdgcn_fmed3f(a[0], 0.0f, inf), __builtin_amdgcn_fmed3f(a[1], 0.0f, inf),
                        __builtin_amdgcn_fmed3f(a[2], 0.0f, inf), __builtin_amdgcn_fmed3f(a[3], 0.0f, inf));
    }
};

// epilogue: raw store to packed LDS (hidden state before normalisation)
struct EpiRawPacked {
    float* dst;
    int lane;
    __device__ __forceinline__ void operator()(int t, const f32x4& a) const {
        reinterpret_cast<float4*>(dst)[t * 64 + lane] = make_float4(a[0], a[1], a[2], a[3]);
    }
};

// epilogue: head logits to LG[e][n] (row-major per env, for the sequential softmax)
struct EpiLogits {
    float* dst;
    int stride, lane;
    __device__ __forceinline__ void operator()(int t, const f32x4& a) const {
        const int q = lane >> 4, e = lane & 15;
        float* p = dst + e * stride + t * 16 + q * 4;
        p[0] = a[0];
        p[1] = a[1];
        p[2] = a[2];
        p[3] = a[3];
    }
};

// copy every layer's padded bias vector into LDS (once per kernel): bias reads then cost an LDS access, not an
// exposed L2 round trip at the head of each layer
__device__ __forceinline__ void stage_biases(const MlpNet& net, float* lds, int tid) {
    // one flat copy (the loads of a thread are independent: they overlap instead of paying an L2 round trip per layer)
    for (int i = tid * 4; i < net.b_count; i += WG_THREADS * 4)
        *reinterpret_cast<float4*>(lds + net.b_base + i) = *reinterpret_cast<const float4*>(net.b_all + i);
}

// normalize_hidden_state (util.py:31-36) for the MLP nets: min/max over the H features of each env, then
// (h - min) / (max - min + 1e-8).  Reads HN (packed), writes HS (packed) and, if gdst != nullptr, the row-major
// hidden state gdst[e*row_stride + k] (float4 stores).  256 threads: 16 per env, each owning chunks of 4 features.
__device__ __forceinline__ void normalize_tile(const MlpNet& net, const float* HN, float* HS, float* const* grow, int tid) {
    const int e = tid >> 4, c0 = tid & 15;
    const int chunks = net.h_pad >> 2;
    float mn = __uint_as_float(0x7f800000u), mx = __uint_as_float(0xff800000u);
    for (int c = c0; c < chunks; c += 16) {
        const int k = c << 2;
        const float* p = HN + pk(k, e);  // pk(k + j, e) = base + j
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (k + j < net.H) {
                const float v = p[j];
                mn = v < mn ? v : mn;
                mx = v > mx ? v : mx;
            }
        }
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        const float omn = __shfl_xor(mn, m, 64), omx = __shfl_xor(mx, m, 64);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
    }
    const float d = (mx - mn) + 1e-8f;
    float* g = grow ? grow[e] : nullptr;
    for (int c = c0; c < chunks; c += 16) {
        const int k = c << 2;
        const int base = pk(k, e);
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float v = (k + j < net.H) ? (HN[base + j] - mn) / d : 0.0f;
            o[j] = v;
            HS[base + j] = v;
        }
        if (g) {
            if (k + 3 < net.H) {
                *reinterpret_cast<float4*>(g + k) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                for (int j = 0; j < 4; j++)
                    if (k + j < net.H) g[k + j] = o[j];
            }
        }
    }
}

// zero-fill a packed buffer region [k_lo, k_hi) x 16 envs (padding rows must be exact zeros)
__device__ __forceinline__ void zero_packed(float* dst, int k_lo, int k_hi, int tid) {
    for (int i = tid; i < (k_hi - k_lo) * 16; i += WG_THREADS) {
        const int k = k_lo + (i >> 4), e = i & 15;
        dst[pk(k, e)] = 0.0f;
    }
}

// reward / value logits rows (LG[head][e][:]) -> scalars OUT[e][head]; 16 lanes per row, 256 threads = 16 rows per pass.
// with_reward == false (initial inference): reward := 0 (network.py:76), only the value rows are reduced.
__device__ __forceinline__ void heads_to_scalars(const MlpNet& net, const MlpLds& o, float* lds, int tid, bool with_reward) {
    const int e = tid >> 4, j = tid & 15;
    const float v = row_logits_to_scalar(lds + o.LG + (16 + e) * o.lg_stride, net.Sv, j);
    float r = 0.0f;
    if (with_reward) r = row_logits_to_scalar(lds + o.LG + e * o.lg_stride, net.Sr, j);
    if (j == 0) {
        lds[o.OUT + e * 4 + 0] = r;
        lds[o.OUT + e * 4 + 1] = v;
    }
}

// ONE-neuron K-split layer (the second layer of an MSE head: value / reward support size 1, network.py:172-182,212-222) on the
// vector ALUs -- a 16-row MFMA tile would carry 15 idle rows.  Thread (e, c, q) of the tile's 256 runs the chain of lane group q of
// quarter c for env e: k = 16g + 4q + i over the quarter's blocks g (ascending), i = 0..3; the chain of (c, q) = (0, 0) starts from
// the bias.  The groups meet as (p0 + p1) + (p2 + p3), the quarters as ((s0 + s1) + s2) + s3 -- the order the CPU restatement
// states for this layer and k_search_fast reproduces from its registers.  Xs: the packed input (pk layout, padding rows zero);
// the packed weights of row 0 sit in lane group q of tile 0: float4 (g * 64 + 16 q).  Writes out[e * stride].
__device__ __forceinline__ void scalar_head_tile(const MlpLayer& L, const float* __restrict__ lds, const float* __restrict__ Xs, float* out, int stride,
                                                 int tid) {
    const int e = tid >> 4, c = (tid >> 2) & 3, q = tid & 3;
    const float4* X4 = reinterpret_cast<const float4*>(Xs);
    const float4* W4 = reinterpret_cast<const float4*>(L.w);
    float p = (c == 0 && q == 0) ? lds[L.b_lds] : 0.0f;
    const int g1 = (c + 1) * L.kq < L.kg ? (c + 1) * L.kq : L.kg;
    for (int g = c * L.kq; g < g1; g++) {
        const float4 x = X4[g * 64 + q * 16 + e], w = ldg4(&W4[g * 64 + q * 16]);
        p = fmaf(x.x, w.x, p); p = fmaf(x.y, w.y, p); p = fmaf(x.z, w.z, p); p = fmaf(x.w, w.w, p);
    }
    float sum = p + __shfl_xor(p, 1, 64);    // p0 + p1 | p2 + p3 (float addition commutes: both lanes of a pair hold the same sum)
    sum = sum + __shfl_xor(sum, 2, 64);      // (p0 + p1) + (p2 + p3)
    const int base = (tid & 63) & ~15;
    const float s0 = __shfl(sum, base, 64), s1 = __shfl(sum, base + 4, 64), s2 = __shfl(sum, base + 8, 64), s3 = __shfl(sum, base + 12, 64);
    if ((tid & 15) == 0) out[e * stride] = ((s0 + s1) + s2) + s3;
}

// dynamics + reward + value for the 16 envs of this tile (network.py:86-111 without the dead policy head unless
// want_policy).  Expects X (packed hidden+onehot) ready in LDS and a barrier already passed.
//   out: lds OUT[e][0] = reward, OUT[e][1] = value ; HS = normalised next hidden ; grow[e] (optional) global rows.
__device__ __forceinline__ void mlp_recurrent_tile(const MlpNet& net, const MlpLds& o, float* lds, float* const* grow, bool want_policy,
                                                   float* pi_out /*LDS [16][A] or null*/, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    gemm_layer(net.L[L_DYN0], lds, lds + o.X, wave, lane, EpiReluPacked{lds + o.H1, lane});
    __syncthreads();
    gemm_layer(net.L[L_DYN1], lds, lds + o.H1, wave, lane, EpiRawPacked{lds + o.HN, lane});
    __syncthreads();
    normalize_tile(net, lds + o.HN, lds + o.HS, grow, tid);
    __syncthreads();
    // reward head reads the UN-normalised state (network.py:195-196), value head the normalised one
    gemm_layer(net.L[L_REW0], lds, lds + o.HN, wave, lane, EpiReluPacked{lds + o.H1, lane});
    gemm_layer(net.L[L_VAL0], lds, lds + o.HS, wave, lane, EpiReluPacked{lds + o.V1, lane});
    __syncthreads();
    if (net.L[L_REW1].n == 1) scalar_head_tile(net.L[L_REW1], lds, lds + o.H1, lds + o.LG, o.lg_stride, tid);
    else gemm_layer(net.L[L_REW1], lds, lds + o.H1, wave, lane, EpiLogits{lds + o.LG, o.lg_stride, lane});
    if (net.L[L_VAL1].n == 1) scalar_head_tile(net.L[L_VAL1], lds, lds + o.V1, lds + o.LG + 16 * o.lg_stride, o.lg_stride, tid);
    else gemm_layer(net.L[L_VAL1], lds, lds + o.V1, (wave + 2) & 3, lane, EpiLogits{lds + o.LG + 16 * o.lg_stride, o.lg_stride, lane});
    __syncthreads();
    heads_to_scalars(net, o, lds, tid, true);
    if (want_policy) {
        __syncthreads();
        gemm_layer(net.L[L_POL0], lds, lds + o.HS, wave, lane, EpiReluPacked{lds + o.H1, lane});
        __syncthreads();
        gemm_layer(net.L[L_POL1], lds, lds + o.H1, wave, lane, EpiLogits{lds + o.LG, o.lg_stride, lane});
        __syncthreads();
        row_softmax(lds + o.LG + (tid >> 4) * o.lg_stride, pi_out + (tid >> 4) * net.A, net.A, tid & 15);
    }
    __syncthreads();
}

// representation + prediction (network.py:62-84).  Expects X (packed obs) in LDS, barrier passed.
//   out: HS = normalised hidden (and grow rows), pi_out LDS [16][A] (softmax), OUT[e][1] = value.
// The root's layers are read once per move, by every workgroup at the same moment, and have long left the L2 by then
// (a move streams ~100 MB of hidden states through it): un-prefetched, each dependent block of the root GEMMs waits a
// full HBM round trip.  Each workgroup touches its share of the lines of `layers` at kernel start -- workgroups are
// dealt round-robin to the 8 XCDs, so the workgroups of one XCD (blockIdx / 8) split that XCD's copy -- and the GEMMs,
// ~25 k cycles later, find them in L2.  Returns a value the caller must keep alive (store it under a condition that is
// never true) so that the loads are not dropped.
__device__ __forceinline__ float prefetch_root_weights(const MlpNet& net, const int* layers, int n_layers, int tid) {
    const int peers = (int)(gridDim.x + 7) >> 3;
    const int parts = peers < 32 ? peers : 32, part = (int)(blockIdx.x >> 3) % parts;
    float acc = 0.0f;
    for (int li = 0; li < n_layers; li++) {
        const MlpLayer& L = net.L[layers[li]];
        const int lines = L.n_tiles * L.kg * 8;  // 1 KiB per (tile, block) = 8 lines of 128 B
        for (int i = part * WG_THREADS + tid; i < lines; i += parts * WG_THREADS) acc += L.w[(size_t)i * 32];  // (a plain load: a non-temporal one would not leave the line in L2)
    }
    return acc;
}

__device__ __forceinline__ void mlp_initial_tile(const MlpNet& net, const MlpLds& o, float* lds, float* const* grow, float* pi_out, int tid,
                                                 bool active = true, bool want_value = true) {
    // `want_value == false`: the search discards the root's value (mcts.py:356-367 expands the root with the prior only), so
    // its value head is not evaluated
    // `active == false`: a wave that only keeps the workgroup barriers uniform (512-thread kernels run this 4-wave
    // pipeline on waves 0-3)
    const int lane = tid & 63, wave = tid >> 6;
    MZ_ROOT_TS_START();
    if (active) gemm_layer(net.L[L_REP0], lds, lds + o.X, wave, lane, EpiReluPacked{lds + o.H1, lane});
    __syncthreads();
    MZ_ROOT_TS(0);
    if (active) gemm_layer(net.L[L_REP1], lds, lds + o.H1, wave, lane, EpiRawPacked{lds + o.HN, lane});
    __syncthreads();
    MZ_ROOT_TS(1);
    if (active) normalize_tile(net, lds + o.HN, lds + o.HS, grow, tid);
    __syncthreads();
    MZ_ROOT_TS(2);
    if (active) {
        gemm_layer(net.L[L_POL0], lds, lds + o.HS, wave, lane, EpiReluPacked{lds + o.H1, lane});
        if (want_value) gemm_layer(net.L[L_VAL0], lds, lds + o.HS, wave, lane, EpiReluPacked{lds + o.V1, lane});
    }
    __syncthreads();
    MZ_ROOT_TS(3);
    // (without the value head the V1 buffer is free: scratch for the policy layer's partial tiles)
    if (want_value || !gemm_one_tile_across_waves(net.L[L_POL1], lds, lds + o.H1, lds + o.V1, wave, lane, active, EpiLogits{lds + o.LG, o.lg_stride, lane})) {
        if (active) {
            gemm_layer(net.L[L_POL1], lds, lds + o.H1, wave, lane, EpiLogits{lds + o.LG, o.lg_stride, lane});
            if (want_value) {
                if (net.L[L_VAL1].n == 1) scalar_head_tile(net.L[L_VAL1], lds, lds + o.V1, lds + o.LG + 16 * o.lg_stride, o.lg_stride, tid);
                else gemm_layer(net.L[L_VAL1], lds, lds + o.V1, (wave + 2) & 3, lane, EpiLogits{lds + o.LG + 16 * o.lg_stride, o.lg_stride, lane});
            }
        }
    }
    __syncthreads();
    MZ_ROOT_TS(4);
    if (active) {
        row_softmax(lds + o.LG + (tid >> 4) * o.lg_stride, pi_out + (tid >> 4) * net.A, net.A, tid & 15);
        if (want_value) heads_to_scalars(net, o, lds, tid, false);
    }
    __syncthreads();
    MZ_ROOT_TS(5);
}

// Fill X with [hidden row (H floats, global; zero padded to h_pad) | one-hot(action) block(s)] for the 16 envs
// (network.py:191-193).  src[e] may be null (env slot unused): zeros.  256 threads: 16 per env, float4 loads.
__device__ __forceinline__ void load_hidden_onehot(const MlpNet& net, float* X, const float* const* src, const int* action, int tid) {
    const int e = tid >> 4, c0 = tid & 15;
    const float* s = src[e];
    const int chunks = net.h_pad >> 2;
    for (int c = c0; c < chunks; c += 16) {
        const int k = c << 2;
        float v[4];
        if (k + 3 < net.H && s) {
            const float4 t = *reinterpret_cast<const float4*>(s + k);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = (k + j < net.H && s) ? s[k + j] : 0.0f;
        }
        *reinterpret_cast<float4*>(X + pk(k, e)) = make_float4(v[0], v[1], v[2], v[3]);
    }
    const int hblocks = net.h_pad >> 4, a_pad = net.x_pad - net.h_pad;
    for (int a = c0; a < a_pad; a += 16) X[pk_act(a, e, hblocks)] = (a < net.A && a == action[e]) ? 1.0f : 0.0f;
}

// Fill X with flattened observations (float32 rows of in_dim).
__device__ __forceinline__ void load_obs(const MlpNet& net, float* X, const float* const* src, int tid) {
    const int e = tid >> 4, c0 = tid & 15;
    const float* s = src[e];
    const int chunks = net.in_pad >> 2;
    for (int c = c0; c < chunks; c += 16) {
        const int k = c << 2;
        const int base = pk(k, e);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int kk = k + j;
            X[base + j] = (s && kk < net.in_dim) ? s[kk] : 0.0f;
        }
    }
}

}  // namespace mz
