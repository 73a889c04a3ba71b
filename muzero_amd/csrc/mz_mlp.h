// mz_mlp.h -- fused MuZero MLP inference for one workgroup tile of 16 environments (gfx950, wave64).
//
// Every Linear layer (network.py:145-149,172-182,212-222) is Y[n][e] = b[n] + sum_k W[n][k] * X[e][k] computed with
// v_mfma_f32_16x16x4_f32: A operand = 16 output neurons x 4 k of the weight matrix, B operand = 4 k x 16 environments
// of the activations, D = 16 neurons x 16 environments.  The MFMA accumulates k in order as one float32 fmaf chain
// (exact f32, no wider internal sum), and the chain's initial value is the bias, so each output equals the oracle's
// sequential fmaf chain bit for bit.  K is never split across waves or accumulators.
//
// Activations live in LDS in "fragment-packed" order so that ONE ds_read_b128 per lane feeds 4 consecutive k-steps:
//     pk(k, e) = ((k >> 4) * 64 + (k & 3) * 16 + e) * 4 + ((k >> 2) & 3)          [float index]
// i.e. lane (q = lane >> 4, e = lane & 15) reads float4 #(g*64 + lane) and gets X[e][16g + 4s + q], s = 0..3 --
// exactly the B operand of k-steps 4g .. 4g+3.  Weights are pre-packed on the host into the mirror-image A-operand
// order, so one global_load_dwordx4 per lane (1 KiB contiguous per wave) feeds the same 4 k-steps:
//     Wp[((t * KG + g) * 64 + lane) * 4 + s] = W[16 t + (lane & 15)][16 g + 4 s + (lane >> 4)]
// Weights are streamed L2 -> VGPR (each element is used once per 16-env tile, so LDS staging would add nothing).
#pragma once
#include "mz_device.h"

namespace mz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE_E = 16;      // environments per workgroup tile
constexpr int WG_THREADS = 256; // 4 waves
constexpr int WG_WAVES = 4;

__device__ __forceinline__ int pk(int k, int e) { return (((k >> 4) * 64 + (k & 3) * 16 + e) << 2) + ((k >> 2) & 3); }

// Layer ids in the packed parameter table (state_dict order, network.py:236-267)
enum { L_REP0 = 0, L_REP1, L_DYN0, L_DYN1, L_REW0, L_REW1, L_POL0, L_POL1, L_VAL0, L_VAL1, L_COUNT };

struct MlpLayer {
    const float* w;  // packed [n_tiles][kg][64][4]
    const float* b;  // [n_tiles*16], zero padded
    int n;           // real output features
    int k;           // real input features
    int n_tiles;     // ceil(n/16)
    int k_steps;     // ceil(k/4)
    int kg;          // ceil(k/16)
    int b_lds;       // float offset of the LDS copy of b (staged once per kernel by stage_biases)
};

struct MlpNet {
    MlpLayer L[L_COUNT];
    int in_dim, A, P, H, Sv, Sr;
    int in_pad, x_pad, p_pad, h_pad;  // multiples of 16: obs, H+A, P, H
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
    const int full = L.k_steps >> 2;
    for (int g = 0; g < full; g++) {
        const float4 x = xp[g * 64];
        float4 w[NACC];
#pragma unroll
        for (int j = 0; j < NACC; j++) w[j] = wp[j][g * 64];
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].x, x.x, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].y, x.y, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].z, x.z, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].w, x.w, acc[j], 0, 0, 0);
    }
    const int rem = L.k_steps & 3;
    if (rem) {
        const int g = full;
        const float4 x = xp[g * 64];
        float4 w[NACC];
#pragma unroll
        for (int j = 0; j < NACC; j++) w[j] = wp[j][g * 64];
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].x, x.x, acc[j], 0, 0, 0);
        if (rem > 1) {
#pragma unroll
            for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].y, x.y, acc[j], 0, 0, 0);
        }
        if (rem > 2) {
#pragma unroll
            for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j].z, x.z, acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < NACC; j++) epi(t0 + j * WG_WAVES, acc[j]);
}

template <typename Epi>
__device__ __forceinline__ void gemm_layer(const MlpLayer& L, const float* __restrict__ lds, const float* __restrict__ Xs, int wave_slot,
                                           int lane, Epi epi) {
    int t = wave_slot;
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
        const int q = lane >> 4, e = lane & 15;
        // pk(16t + 4q + r, e) = ((t*64 + r*16 + e) << 2) + q
        float* p = dst + ((t * 64 + e) << 2) + q;
        // ReLU as one v_med3_f32 (median of x, 0, +inf == max(x, 0) for every non-NaN x)
        const float inf = __uint_as_float(0x7f800000u);
        p[0 * 64] = __builtin_amdgcn_fmed3f(a[0], 0.0f, inf);
        p[1 * 64] = __builtin_amdgcn_fmed3f(a[1], 0.0f, inf);
        p[2 * 64] = __builtin_amdgcn_fmed3f(a[2], 0.0f, inf);
        p[3 * 64] = __builtin_amdgcn_fmed3f(a[3], 0.0f, inf);
    }
};

// epilogue: raw store to packed LDS (hidden state before normalisation)
struct EpiRawPacked {
    float* dst;
    int lane;
    __device__ __forceinline__ void operator()(int t, const f32x4& a) const {
        const int q = lane >> 4, e = lane & 15;
        float* p = dst + ((t * 64 + e) << 2) + q;
        p[0 * 64] = a[0];
        p[1 * 64] = a[1];
        p[2 * 64] = a[2];
        p[3 * 64] = a[3];
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
    for (int l = 0; l < L_COUNT; l++) {
        const MlpLayer& L = net.L[l];
        for (int i = tid; i < L.n_tiles * 16; i += WG_THREADS) lds[L.b_lds + i] = L.b[i];
    }
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
        const float* p = HN + (((k >> 4) * 64 + e) << 2) + ((k >> 2) & 3);  // pk(k + j, e) = base + j*64
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (k + j < net.H) {
                const float v = p[j * 64];
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
        const int base = (((k >> 4) * 64 + e) << 2) + ((k >> 2) & 3);
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float v = (k + j < net.H) ? (HN[base + j * 64] - mn) / d : 0.0f;
            o[j] = v;
            HS[base + j * 64] = v;
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
    gemm_layer(net.L[L_REW1], lds, lds + o.H1, wave, lane, EpiLogits{lds + o.LG, o.lg_stride, lane});
    gemm_layer(net.L[L_VAL1], lds, lds + o.V1, (wave + 2) & 3, lane, EpiLogits{lds + o.LG + 16 * o.lg_stride, o.lg_stride, lane});
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
__device__ __forceinline__ void mlp_initial_tile(const MlpNet& net, const MlpLds& o, float* lds, float* const* grow, float* pi_out, int tid,
                                                 bool active = true) {
    // `active == false`: a wave that only keeps the workgroup barriers uniform (512-thread kernels run this 4-wave
    // pipeline on waves 0-3)
    const int lane = tid & 63, wave = tid >> 6;
    if (active) gemm_layer(net.L[L_REP0], lds, lds + o.X, wave, lane, EpiReluPacked{lds + o.H1, lane});
    __syncthreads();
    if (active) gemm_layer(net.L[L_REP1], lds, lds + o.H1, wave, lane, EpiRawPacked{lds + o.HN, lane});
    __syncthreads();
    if (active) normalize_tile(net, lds + o.HN, lds + o.HS, grow, tid);
    __syncthreads();
    if (active) {
        gemm_layer(net.L[L_POL0], lds, lds + o.HS, wave, lane, EpiReluPacked{lds + o.H1, lane});
        gemm_layer(net.L[L_VAL0], lds, lds + o.HS, wave, lane, EpiReluPacked{lds + o.V1, lane});
    }
    __syncthreads();
    if (active) {
        gemm_layer(net.L[L_POL1], lds, lds + o.H1, wave, lane, EpiLogits{lds + o.LG, o.lg_stride, lane});
        gemm_layer(net.L[L_VAL1], lds, lds + o.V1, (wave + 2) & 3, lane, EpiLogits{lds + o.LG + 16 * o.lg_stride, o.lg_stride, lane});
    }
    __syncthreads();
    if (active) {
        row_softmax(lds + o.LG + (tid >> 4) * o.lg_stride, pi_out + (tid >> 4) * net.A, net.A, tid & 15);
        heads_to_scalars(net, o, lds, tid, false);
    }
    __syncthreads();
}

// Fill X with [hidden row (H floats, global) | one-hot(action)] for the 16 envs (network.py:191-193).
// src[e] may be null (env slot unused): zeros.  256 threads: 16 per env, float4 loads.
__device__ __forceinline__ void load_hidden_onehot(const MlpNet& net, float* X, const float* const* src, const int* action, int tid) {
    const int e = tid >> 4, c0 = tid & 15;
    const float* s = src[e];
    const int chunks = net.x_pad >> 2;
    for (int c = c0; c < chunks; c += 16) {
        const int k = c << 2;
        float v[4];
        if (k + 3 < net.H && s) {
            const float4 t = *reinterpret_cast<const float4*>(s + k);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int kk = k + j;
                float t = 0.0f;
                if (kk < net.H) t = s ? s[kk] : 0.0f;
                else if (kk < net.H + net.A) t = (kk - net.H == action[e]) ? 1.0f : 0.0f;
                v[j] = t;
            }
        }
        const int base = (((k >> 4) * 64 + e) << 2) + ((k >> 2) & 3);
#pragma unroll
        for (int j = 0; j < 4; j++) X[base + j * 64] = v[j];
    }
}

// Fill X with flattened observations (float32 rows of in_dim).
__device__ __forceinline__ void load_obs(const MlpNet& net, float* X, const float* const* src, int tid) {
    const int e = tid >> 4, c0 = tid & 15;
    const float* s = src[e];
    const int chunks = net.in_pad >> 2;
    for (int c = c0; c < chunks; c += 16) {
        const int k = c << 2;
        const int base = (((k >> 4) * 64 + e) << 2) + ((k >> 2) & 3);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int kk = k + j;
            X[base + j * 64] = (s && kk < net.in_dim) ? s[kk] : 0.0f;
        }
    }
}

}  // namespace mz
