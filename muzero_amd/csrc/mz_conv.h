// mz_conv.h -- conv-tower inference kernels (MuZeroBoardGameNet network.py:540-574, MuZeroAtariNet :501-537) for gfx950.
//
// k_conv3x3<NPT>: 3x3 convolution (stride 1 or 2, pad 1) with eval-mode BatchNorm folded into weight/bias, optional
// residual add and ReLU (ResNetBlock network.py:293-299), as an implicit GEMM on v_mfma_f32_16x16x4_f32:
//     D[co][pixel] = bias[co] + sum_k W[co][k] * X[k][pixel],   k = (16-channel block, tap (ky,kx), channel in block)
// A operand = 16 output channels x 4 k of the weights (pre-packed fragment order, streamed from L2), B operand =
// 4 channels x 16 output pixels read from an LDS-staged input slab (16 channels x tile-with-halo).  The k order is ONE
// fmaf chain per output in exactly the oracle's order, so results equal the oracle bit for bit; zero padding contributes
// fma(w, 0, acc) == acc.
//   workgroup = 256 threads = one image (blockIdx.y) x one spatial tile of up to NPT*16 output pixels (blockIdx.x) x one
//   slice of 128 output channels (blockIdx.z; wave w owns channel tiles w and w + 4 of the slice); accumulators:
//   2 x NPT tiles of 16x16 per wave.
// The dynamics net's action planes (network.py:440-444: element f = c*h*w + y*w + x of the [A,h,w] block is 1 iff
// f % A == action) are generated while staging, never materialised.
#pragma once
#include "mz_mlp.h"

namespace mz {

struct ConvLaunch {
    // input: per-image base pointers (gather from the node store) or a dense buffer
    const float* const* in_ptrs;  // [B] or null
    const float* in;              // dense [B][cin_real][ih][iw] if in_ptrs == null
    const int* action;            // [B] or null: channels >= cin_real are action planes over num_actions
    int num_actions;
    int cin_real;                 // channels present in memory
    int cin;                      // logical input channels (cin_real + num_actions planes), k runs over pad16(cin)
    int ih, iw, oh, ow, stride;
    int cout;                     // output channels; blockIdx.z selects a slice of 128 (8 tiles of 16)
    const float* w;               // packed [co_tile][cb][tap][64 lanes][4]
    const float* bias;            // [pad16(cout)]
    const float* residual;        // dense [B][cout][oh][ow] or null
    float* out;                   // dense [B][cout][oh][ow]
    int relu;
    int th, tw;                   // spatial tile (th*tw <= NPT*16)
    int tiles_x, tiles_y;
    int B;
};

template <int NPT>
__global__ __launch_bounds__(256) void k_conv3x3(const ConvLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* slab = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, j = lane & 15;
    const int b = blockIdx.y;
    const int tile = blockIdx.x, ty0 = (tile / L.tiles_x) * L.th, tx0 = (tile % L.tiles_x) * L.tw;
    const int sih = (L.th - 1) * L.stride + 3, siw = (L.tw - 1) * L.stride + 3, plane = sih * siw;
    const float* src = L.in_ptrs ? L.in_ptrs[b] : L.in + (size_t)b * L.cin_real * L.ih * L.iw;
    const int act = L.action ? L.action[b] : 0;
    const int co_tiles = (L.cout + 15) >> 4, n_cb = (L.cin + 15) >> 4;
    // this lane's output pixels: slot p = pt*16 + j -> (py, px) inside the tile; slab offset of its top-left tap
    int off[NPT];
    bool pv[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; pt++) {
        const int p = pt * 16 + j, py = p / L.tw, px = p - py * L.tw;
        pv[pt] = (p < L.th * L.tw) && (ty0 + py < L.oh) && (tx0 + px < L.ow);
        off[pt] = pv[pt] ? (py * L.stride) * siw + px * L.stride : 0;
    }
    f32x4 acc[2][NPT];
    int cot[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        cot[c] = blockIdx.z * 8 + wave + 4 * c;
        const bool ok = cot[c] < co_tiles;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) bv = *reinterpret_cast<const float4*>(L.bias + cot[c] * 16 + q * 4);
#pragma unroll
        for (int pt = 0; pt < NPT; pt++) acc[c][pt] = f32x4{bv.x, bv.y, bv.z, bv.w};
    }
    const int iy0 = ty0 * L.stride - 1, ix0 = tx0 * L.stride - 1;
    for (int cb = 0; cb < n_cb; cb++) {
        __syncthreads();
        for (int i = tid; i < 16 * plane; i += 256) {
            const int c = i / plane, r = i - c * plane, sy = r / siw, sx = r - sy * siw;
            const int ch = cb * 16 + c, gy = iy0 + sy, gx = ix0 + sx;
            float v = 0.0f;
            if (ch < L.cin && gy >= 0 && gy < L.ih && gx >= 0 && gx < L.iw) {
                if (ch < L.cin_real) v = src[((size_t)ch * L.ih + gy) * L.iw + gx];
                else v = ((((ch - L.cin_real) * L.ih + gy) * L.iw + gx) % L.num_actions == act) ? 1.0f : 0.0f;
            }
            slab[i] = v;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int toff = (tap / 3) * siw + (tap % 3);
            float4 wv[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                wv[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cot[c] < co_tiles) wv[c] = reinterpret_cast<const float4*>(L.w)[(((size_t)cot[c] * n_cb + cb) * 9 + tap) * 64 + lane];
            }
#pragma unroll
            for (int pt = 0; pt < NPT; pt++) {
                const float* xp = slab + q * plane + off[pt] + toff;
                const float x0 = xp[0], x1 = xp[4 * plane], x2 = xp[8 * plane], x3 = xp[12 * plane];
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    if (cot[c] < co_tiles) {
                        acc[c][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c].x, x0, acc[c][pt], 0, 0, 0);
                        acc[c][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c].y, x1, acc[c][pt], 0, 0, 0);
                        acc[c][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c].z, x2, acc[c][pt], 0, 0, 0);
                        acc[c][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c].w, x3, acc[c][pt], 0, 0, 0);
                    }
                }
            }
        }
    }
    // epilogue: D row 4q + r = output channel inside the tile, column j = pixel slot
#pragma unroll
    for (int c = 0; c < 2; c++) {
        if (cot[c] >= co_tiles) continue;
#pragma unroll
        for (int pt = 0; pt < NPT; pt++) {
            if (!pv[pt]) continue;
            const int p = pt * 16 + j, py = p / L.tw, px = p - py * L.tw;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int co = cot[c] * 16 + q * 4 + r;
                if (co < L.cout) {
                    const size_t o = (((size_t)b * L.cout + co) * L.oh + ty0 + py) * L.ow + tx0 + px;
                    float v = acc[c][pt][r];
                    if (L.residual) v = v + L.residual[o];
                    if (L.relu && !(v > 0.0f)) v = 0.0f;
                    L.out[o] = v;
                }
            }
        }
    }
}

// nn.AvgPool2d(3, 2, 1), count_include_pad (network.py:337,342): taps summed row-major, then / 9
__global__ void k_avgpool(const float* in, float* out, int B, int C, int ih, int iw, int oh, int ow) {
    const size_t n = (size_t)B * C * oh * ow;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = i % ow, oy = (i / ow) % oh;
        const size_t bc = i / ((size_t)ow * oh);
        const float* p = in + bc * ih * iw;
        float acc = 0.0f;
        for (int ky = 0; ky < 3; ky++)
            for (int kx = 0; kx < 3; kx++) {
                const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
                if (iy >= 0 && iy < ih && ix >= 0 && ix < iw) acc = acc + p[iy * iw + ix];
            }
        out[i] = acc / 9.0f;
    }
}

// normalize_hidden_state for conv states (util.py:31-36): min/max over the channels of each pixel.
// in dense [B][C][hw]; written to every non-null destination: out_ptrs[b] (node store rows), dense out_a, dense out_b.
__global__ void k_normalize_planes(const float* in, float* const* out_ptrs, float* out_a, float* out_b, int B, int C, int hw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * hw) return;
    const int b = i / hw, p = i - b * hw;
    const float* s = in + (size_t)b * C * hw + p;
    float mn = s[0], mx = s[0];
    for (int c = 1; c < C; c++) {
        const float v = s[(size_t)c * hw];
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
    const float d = (mx - mn) + 1e-8f;
    float* o0 = out_ptrs ? out_ptrs[b] + p : nullptr;
    float* o1 = out_a ? out_a + (size_t)b * C * hw + p : nullptr;
    float* o2 = out_b ? out_b + (size_t)b * C * hw + p : nullptr;
    for (int c = 0; c < C; c++) {
        const float v = (s[(size_t)c * hw] - mn) / d;
        if (o0) o0[(size_t)c * hw] = v;
        if (o1) o1[(size_t)c * hw] = v;
        if (o2) o2[(size_t)c * hw] = v;
    }
}

// Head (network.py:424-430, 472-486): 1x1 conv (C -> oc planes, BN folded) + ReLU + flatten + Linear(oc*hw -> n_out),
// then for value/reward heads softmax-expectation-transform (util.py:70-93) or, for the policy head, softmax.
// One workgroup of 256 threads per image; chains in the oracle's order.
struct HeadLaunch {
    const float* in;       // dense [B][C][hw]
    const float* const* in_ptrs;  // or per-image pointers
    int C, hw, oc, n_out;
    const float* cw;       // [oc][C]  (1x1 conv, BN folded)
    const float* cb;       // [oc]
    const float* lw;       // [n_out][oc*hw]
    const float* lb;       // [n_out]
    int mode;              // 0: scalar via logits_to_transformed_expected_value (n_out == 1: identity); 1: softmax -> probs
    float* out_scalar;     // [B]
    float* out_probs;      // [B][n_out]
    int B;
};

__global__ __launch_bounds__(256) void k_head(const HeadLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* feat = reinterpret_cast<float*>(smem);     // [oc*hw]
    float* lg = feat + ((L.oc * L.hw + 3) & ~3);      // [n_out]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* src = L.in_ptrs ? L.in_ptrs[b] : L.in + (size_t)b * L.C * L.hw;
    for (int i = tid; i < L.oc * L.hw; i += 256) {
        const int o = i / L.hw, p = i - o * L.hw;
        float acc = L.cb[o];
        for (int c = 0; c < L.C; c++) acc = fmaf(src[(size_t)c * L.hw + p], L.cw[o * L.C + c], acc);
        feat[i] = acc > 0.0f ? acc : 0.0f;
    }
    __syncthreads();
    const int K = L.oc * L.hw;
    for (int n = tid; n < L.n_out; n += 256) {
        float acc = L.lb[n];
        const float* w = L.lw + (size_t)n * K;
        for (int k = 0; k < K; k++) acc = fmaf(feat[k], w[k], acc);
        lg[n] = acc;
    }
    __syncthreads();
    if (tid < 16) {  // one 16-lane row (DPP butterflies need the whole row active)
        if (L.mode == 0) {
            const float v = row_logits_to_scalar(lg, L.n_out, tid);
            if (tid == 0) L.out_scalar[b] = v;
        } else {
            row_softmax(lg, L.out_probs + (size_t)b * L.n_out, L.n_out, tid);
        }
    }
}

}  // namespace mz
