// mz_learn_conv.h -- learner-step kernels of the CONV nets (MuZeroBoardGameNet, network.py:540-574) for gfx950: row f2 of SURVEY 8, second half.
//
// What one update computes (pipeline.py:541-612 `calc_loss` on network.py:273-299 ResNetBlock, :396-446 DynamicsConvNet, :449-498 PredictionConvNet,
// :356-393 RepresentationConvNet, all BatchNorm2d layers in TRAIN mode: batch statistics, running-stat update), then backward:
//
//   k_lc_gather        replay rows -> dense float32 observations [B][C0][hw], actions [K][B]
//   k_lc_conv<NPT>     3x3 convolution as implicit GEMM on v_mfma_f32_16x16x4_f32, used for BOTH directions:
//                        forward   y = conv(f(x))         f applied while staging: identity | relu(a x + b) (the previous layer's BatchNorm + ReLU,
//                                                          never materialised) | relu(a y2 + b + x) (the previous BLOCK's output, written through once) |
//                                                          action planes generated on the fly (network.py:440-444)
//                                  epilogue: per-channel partial sums of the batch statistics, taken around a pivot: (p, sum (y - p), sum (y - p)^2, n)
//                        dgrad     g = convT(dy)          dy = c1 dz + c2 y + c3 (BatchNorm backward, applied while staging from TWO tensors),
//                                                          weights = the transposed / tap-flipped packed copy
//                                  epilogue: + skip gradient, ReLU mask of the layer below, partial sums (sum dz, sum dz y) of ITS BatchNorm backward
//   k_lc_wgrad         dW[co][ci][tap] = sum_{b,p} dy[b][co][p] x[b][ci][p + tap]: MFMA with the reduction over PIXELS; both operands staged through
//                      LDS in a zero-padded row-pitch layout so that a tap is a register choice, never a re-read; partials per image chunk
//   k_lc_wgrad_act     the same for the action planes of the dynamics net's first conv (one-hot-like planes: a gather of dy), board games
//   k_lc_wreduce       chunk partials -> gradient (torch layout), fixed order
//   k_lc_bn_fwd / k_lc_bn_bwd   finalize the partial sums -> per-channel coefficients (a, b | c1, c2, c3), running statistics, dgamma / dbeta
//   k_lc_apply         block output x' = relu(a y + b + x) (the residual needs it materialised)
//   k_lc_entry         gradient entering a tower: [normalize_hidden_state backward (util.py:31-36) of the next step's gradient, x 0.5 (pipeline.py:584)]
//                      + head gradient, ReLU mask, BatchNorm-backward partial sums
//   k_lc_normalize     min / max normalisation over the channels of each pixel (forward)
//   k_lch_*            the three heads (1x1 conv + BatchNorm + ReLU + Linear, network.py:424-430,472-486): forward, losses (pipeline.py:586-597:
//                      squared error | soft-target cross entropy, importance weights, 1/K gradient scale :600), priorities (:603-609), backward
//   k_lc_sqsum / k_lc_adam / k_lc_pack   clip_grad_norm_ + torch.optim.Adam (L2 decay in the gradient) on the flat vector, operand re-pack
//
// No atomics: every reduction has a fixed order, an update is bit-reproducible.  fp32 throughout (the reference's arithmetic); parity with the
// reference / autograd is by tolerance (tests/test_gpu_conv_learner.py), summation orders are this file's own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mz_device.h"

namespace mzlc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum { IN_IDENT = 0, IN_BNRELU = 1, IN_BNBWD = 2, IN_BNRES = 3 };  // IN_BNRES: relu(c1 x0 + c2 x1 + c3) -- a block output relu(a y2 + b + x) formed while staging
enum { ST_NONE = 0, ST_FWD = 1, ST_BWD = 2 };

__device__ __forceinline__ int lc_idiv(int p, float rcp_d) { return (int)(((float)p + 0.5f) * rcp_d); }  // exact for 0 <= p < 4096 (index arithmetic)
__device__ __forceinline__ float4 ld4(const __amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
}
__device__ __forceinline__ float ld1(const __amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mkrs(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000); }

// two independent jobs of the same kind side by side in one launch (blockIdx.y < na: job a): the prediction and dynamics towers of one unroll
// step have no dependency on each other, and a batch-128 conv launch alone is one workgroup per CU
template <typename J>
struct Pair {
    J a, b;
    int na;  // workgroup rows (blockIdx.y) of job a; rows of job b follow
    int remap;  // k_lc_wgrad: XCD-aware placement of the workgroups (see there); 0: the launch order
};

// ---------------------------------------------------------------------------------------------------------------------------------
// 3x3 convolution, whole images per workgroup.  workgroup = 256 threads = G images (blockIdx.y) x 64 output channels (blockIdx.z; wave w owns
// channel tile 4 z + w) x all pixels (NPT tiles of 16 pixel slots).  Per 16-channel input block: the slab (images + zero halo,
// [slot = channel & 3][position][channel >> 2]) is double-buffered in LDS, the next block fetched global -> registers between the MFMAs of
// this one, transformed and written behind the tap loop; weights stream out of L2 through a register ring (1 KiB per (block, tap) and wave).
// ---------------------------------------------------------------------------------------------------------------------------------
struct LcConv {
    const float* in0;      // [B][cin_real][hw]: x | y (IN_BNRELU) | dz (IN_BNBWD)
    const float* in1;      // IN_BNBWD: y; IN_BNRES: the residual x
    float* mat_out;        // IN_BNRES / IN_BNRELU: the staged (transformed) input is also WRITTEN here by the channel-slice-0 workgroups (the block output the
                           // residual, the weight gradient and the masks need materialised), or null
    const float* coef;     // [3][cpad_in]: (a, b, -) | (c1, c2, c3)
    const int* action;     // [B] or null: channels cin_real .. cin - 1 are the action planes
    const float* w;        // packed [co_tiles][n_cb][9][64][4]: element [lane = (q, j)][i] = W[co = 16 ct + j][ci = 16 cb + 4 i + q][tap]
    float* out;            // [B][cout][hw]
    const float* skip;     // epilogue: added to the result (residual-path gradient), or null
    const float* mask;     // epilogue: result zeroed where (ma * mask + mb) <= 0 (mcoef == null: mask <= 0), or null
    const float* mcoef;    // [2][cpad_out] or null
    const float* partner;  // ST_BWD: y of the layer whose BatchNorm backward the partial sums feed
    float* stat_part;      // [groups][cpad_out][2]
    int in_mode, stat_mode;
    int num_actions, cin_real, cin, n_cb, cpad_in;
    int cout, co_tiles, cpad_out;
    int B, G, h, w_img;
    int qstride;           // LDS floats per slot plane (>= 4 * G * (h + 2) * (w + 2), multiple of 64)
    int tapmask;           // 0 / 0x1ff: all nine taps; else the taps this conv has (a parity plane of a stride-2 conv: 1, 2 or 4 of them), weights packed compactly
    int halo_in;           // round 6, the tile path: 1: in0 holds (h + 2) x (w + 2) positions per image and channel -- the h x w tile WITH its one-pixel halo
                           // (k_lc_tile_gather) -- which are staged straight onto the slab, ring included; the outputs are the h x w INNER
                           // positions only.  (Before: the haloed tile was convolved as an image of its own, (h + 2) x (w + 2) outputs of which the
                           // scatter kept h x w -- 252 of 192 at the 48 x 48 stage, 196 of 144 at 24 x 24: a quarter of the MFMAs thrown away.)
                           // IN_IDENT staging, G == 1, no action planes.
    int out_plane;         // round 6, with halo_in: 1: the "images" are the tiles of planes [B][cout][pl_h][pl_w] cut pl_nty x pl_ntx (tile = h x w_img inner
    int pl_h, pl_w, pl_nty, pl_ntx;  // positions) and `out` / `skip` ARE those planes: the epilogue writes the tile's outputs where they belong (and adds the
                           // skip from there) -- k_lc_tile_scatter's job, without the pass over the tiles; ST_FWD statistics per tile as for any image.
};

#ifndef MZLC_EPI_PIPE
#define MZLC_EPI_PIPE 1
#endif
// taps of a TAPMASK build: their number and the k-th one (compile-time)
constexpr int tm_count(int m) { int c = 0; for (int i = 0; i < 9; i++) c += (m >> i) & 1; return c; }
constexpr int tm_tap(int m, int k) { for (int i = 0; i < 9; i++) if ((m >> i) & 1) { if (k == 0) return i; k--; } return 0; }

// MODE: the staging transform (IN_*) as a compile-time fact; SIDE > 0: a SIDE x SIDE board, one image per workgroup, as compile-time constants
// (the launcher checks them) -- index arithmetic, tap offsets and predicates become immediates.
// TAPMASK != 0x1ff: only the taps of the mask exist (the parity planes of the Atari representation's stride-2 convs have 1, 2, 2 and 4 of the
// nine: run with all nine and zero weights they cost four convolutions for one); their weights are packed [co tile][block][tap of the mask] and
// come one block ahead through two register sets instead of the ring.
// PLANE: the out_plane epilogue (the tile path's stride-1 convs) as a build of its own -- the towers' builds do not carry its address arithmetic
// (measured: as a run-time branch it cost the 19 x 19 net's update 0.3 ms of 45.5).
template <int NPT, int MODE, int SIDE, int TAPMASK = 0x1ff, bool PLANE = false>
__global__ __launch_bounds__(256, 2) void k_lc_conv(const Pair<LcConv> PJ) {
    const bool second = (int)blockIdx.y >= PJ.na;
    LcConv L = second ? PJ.b : PJ.a;
    if constexpr (SIDE > 0) {
        L.h = SIDE; L.w_img = SIDE; L.G = 1; L.qstride = (4 * (SIDE + 2) * (SIDE + 2) + 63) & ~63;
    }
    L.in_mode = MODE;
    const int by = second ? (int)blockIdx.y - PJ.na : (int)blockIdx.y;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* slab = reinterpret_cast<float*>(smem);
    const int bufsz = 4 * L.qstride;
    float* s_coef = slab + 2 * bufsz;  // [3][cpad_in]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
    const int hw = L.h * L.w_img, siw = L.w_img + 2, plane = (L.h + 2) * siw;
    const int img0 = by * L.G;
    const int QP = (hw + 3) >> 2;
    const float r_qp = 1.0f / (float)QP, r_iw = 1.0f / (float)L.w_img, r_hw = 1.0f / (float)hw;
    const __amdgpu_buffer_rsrc_t rs_in0 = mkrs(L.in0), rs_in1 = mkrs(L.in1 ? L.in1 : L.in0), rs_mat = mkrs(L.mat_out ? L.mat_out : L.in0);
    // ---- staging plan: lane t of every wave owns pixel quad t of the group; wave w the channels {w, 4 + w, 8 + w, 12 + w} of each block ----
    // (halo_in: the input "image" is the slab itself -- `plane` positions per channel, ring included, in the slab's own row pitch -- so a lane's
    // quad is four consecutive slab positions and the staging is a linear copy; one image per workgroup)
    const bool halo_in = L.halo_in != 0;
    const int in_hw = halo_in ? plane : hw;                 // positions per image and channel of in0
    const int QPs = halo_in ? (plane + 3) >> 2 : QP;
    const float r_qps = halo_in ? 1.0f / (float)QPs : r_qp;
    const int sg = lc_idiv(lane, r_qps), qd = lane - sg * QPs, p0 = qd * 4, bimg = img0 + sg;
    const bool w_ok = lane < L.G * QPs && bimg < L.B;
    const int cimg = bimg < L.B ? bimg : L.B - 1;
    const unsigned w_voff = (unsigned)(((size_t)cimg * L.cin_real * in_hw + (size_t)(w_ok ? p0 : 0)) * sizeof(float));
    const int w_act = (w_ok && L.action) ? L.action[cimg] : -1;
    int w_spos[4], w_pm[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int pp = p0 + e, py = lc_idiv(pp, r_iw), px = pp - py * L.w_img;
        w_spos[e] = (w_ok && pp < in_hw) ? ((sg < L.G ? sg : 0) * plane + (halo_in ? pp : (py + 1) * siw + px + 1)) * 4 + wave * L.qstride : -1;
        w_pm[e] = L.cin > L.cin_real ? pp % L.num_actions : 0;
    }
    float4 sv0[4], sv1[4];  // [i]: channel 4 i + wave of the block in flight, pixels p0 .. p0 + 3
    auto fetch = [&](int cb, int i) {
        const int ch = cb * 16 + 4 * i + wave, chc = ch < L.cin_real ? ch : 0;
        sv0[i] = ld4(rs_in0, w_voff, chc * in_hw * (int)sizeof(float));
        if (L.in_mode == IN_BNBWD || L.in_mode == IN_BNRES) sv1[i] = ld4(rs_in1, w_voff, chc * in_hw * (int)sizeof(float));
    };
    auto transform_store = [&](int cb, int buf) {
        float v[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int ch = cb * 16 + 4 * i + wave;  // wave-uniform
            const float x[4] = {sv0[i].x, sv0[i].y, sv0[i].z, sv0[i].w};
            if (ch < L.cin_real) {
                if (L.in_mode == IN_BNRELU) {
                    const float a = s_coef[ch], b = s_coef[L.cpad_in + ch];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float t = fmaf(a, x[e], b);
                        v[i][e] = t > 0.0f ? t : 0.0f;
                    }
                } else if (L.in_mode == IN_BNBWD) {
                    const float c1 = s_coef[ch], c2 = s_coef[L.cpad_in + ch], c3 = s_coef[2 * L.cpad_in + ch];
                    const float y[4] = {sv1[i].x, sv1[i].y, sv1[i].z, sv1[i].w};
#pragma unroll
                    for (int e = 0; e < 4; e++) v[i][e] = fmaf(c1, x[e], fmaf(c2, y[e], c3));
                } else if (L.in_mode == IN_BNRES) {  // relu(a y2 + b + x): the op order of k_lc_apply (same bits)
                    const float a = s_coef[ch], b = s_coef[L.cpad_in + ch];
                    const float r[4] = {sv1[i].x, sv1[i].y, sv1[i].z, sv1[i].w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float t = fmaf(a, x[e], b) + r[e];
                        v[i][e] = t > 0.0f ? t : 0.0f;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[i][e] = x[e];
                }
            } else {  // action planes (network.py:440-444: flat element f = c * hw + pixel of the [A, h, w] block is 1 iff f % A == action), or padding
                const int t = ch < L.cin ? (int)(((long long)(ch - L.cin_real) * hw) % L.num_actions) : 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    int m = w_pm[e] + t;
                    m = m >= L.num_actions ? m - L.num_actions : m;
                    v[i][e] = (ch < L.cin && m == w_act) ? 1.0f : 0.0f;
                }
            }
        }
        float* d = slab + buf * bufsz;
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (w_spos[e] >= 0) *reinterpret_cast<float4*>(d + w_spos[e]) = make_float4(v[0][e], v[1][e], v[2][e], v[3][e]);
        if ((MODE == IN_BNRES || MODE == IN_BNRELU) && L.mat_out && blockIdx.z == 0 && w_ok) {  // write-through: every (image, channel, pixel) is staged once per slice
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int ch = cb * 16 + 4 * i + wave;
                if (ch < L.cin_real) {
                    float* o = L.mat_out + ((size_t)cimg * L.cin_real + ch) * hw + p0;
                    if (p0 + 3 < hw) {  // (dword-aligned 16-byte store, like the epilogue's)
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[i][0]), __float_as_uint(v[i][1]), __float_as_uint(v[i][2]), __float_as_uint(v[i][3])},
                                                               rs_mat, w_voff, ch * hw * (int)sizeof(float), 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (p0 + e < hw) o[e] = v[i][e];
                    }
                }
            }
        }
    };
    // ---- A-operand rows of this lane: pixel slot p = pt * 16 + j -> image g of the group, pixel (py, px) ----
    int off[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; pt++) {
        const int p = pt * 16 + j, g = lc_idiv(p, r_hw), pp = p - g * hw, py = lc_idiv(pp, r_iw), px = pp - py * L.w_img;
        off[pt] = (g < L.G ? g * plane + py * siw + px : 0) * 4 + q * L.qstride;
    }
    f32x4 acc[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; pt++) acc[pt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int cot = blockIdx.z * 4 + wave, ctc = cot < L.co_tiles ? cot : L.co_tiles - 1;
    const __amdgpu_buffer_rsrc_t rs_w = mkrs(L.w);
    constexpr int NTAP = TAPMASK == 0x1ff ? 9 : tm_count(TAPMASK);
    const int wbase = ctc * L.n_cb * NTAP * 1024, n_steps = L.n_cb * NTAP;
    auto wload = [&](int step) {
        const int sc = step < n_steps ? step : n_steps - 1;
        return ld4(rs_w, lane * 16, wbase + sc * 1024);
    };
    constexpr int WD = (NPT <= 9) ? 9 : 3;
    float4 wr[WD];
    float4 wc[NTAP < 9 ? NTAP : 1], wn[NTAP < 9 ? NTAP : 1];  // TAPMASK builds: this block's and the next block's weights
    // the first weights and the first slab are requested BEFORE the LDS fill below: their latency runs under it
    if constexpr (NTAP == 9) {
#pragma unroll
        for (int s0 = 0; s0 < WD - 1; s0++) wr[s0] = wload(s0);
    } else {
#pragma unroll
        for (int i = 0; i < NTAP; i++) wc[i] = wload(i);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) fetch(0, i);
    for (int i = tid; i < bufsz / 2; i += 256) reinterpret_cast<float4*>(slab)[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // both buffers, halo included
    if (L.in_mode != IN_IDENT)
        for (int i = tid; i < 3 * L.cpad_in; i += 256) s_coef[i] = L.coef[i];
    __syncthreads();
    transform_store(0, 0);
    __syncthreads();
    float4 xr[3];
    const int FL = (L.in_mode == IN_BNBWD || L.in_mode == IN_BNRES) ? 2 : 1;  // staging loads per tap (taps 0..3)
    for (int cb = 0; cb < L.n_cb; cb++) {
        const float* sb = slab + (cb & 1) * bufsz;
        if constexpr (NTAP < 9) {
            const int cbn = cb + 1 < L.n_cb ? cb + 1 : cb;
#pragma unroll
            for (int i = 0; i < NTAP; i++) wn[i] = wload(cbn * NTAP + i);
#pragma unroll
            for (int i = 0; i < 4; i++) fetch(cbn, i);
            constexpr int t0 = tm_tap(TAPMASK, 0), t1 = tm_tap(TAPMASK, NPT > 1 ? 0 : (NTAP > 1 ? 1 : 0));
            xr[0] = *reinterpret_cast<const float4*>(sb + off[0] + ((t0 / 3) * siw + (t0 % 3)) * 4);
            xr[1] = *reinterpret_cast<const float4*>(sb + off[NPT > 1 ? 1 : 0] + ((t1 / 3) * siw + (t1 % 3)) * 4);
#pragma unroll
            for (int ti = 0; ti < NTAP; ti++) {
#pragma unroll
                for (int pt = 0; pt < NPT; pt++) {
                    const int n = ti * NPT + pt, n2 = n + 2;
                    if (n2 < NTAP * NPT) {
                        const int ti2 = n2 / NPT, pt2 = n2 - ti2 * NPT, tap2 = tm_tap(TAPMASK, ti2);
                        xr[n2 % 3] = *reinterpret_cast<const float4*>(sb + off[pt2] + ((tap2 / 3) * siw + (tap2 % 3)) * 4);
                    }
                    const float4 x4 = xr[n % 3], w4 = wc[ti];
                    acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.x, w4.x, acc[pt], 0, 0, 0);
                    acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.y, w4.y, acc[pt], 0, 0, 0);
                    acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.z, w4.z, acc[pt], 0, 0, 0);
                    acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.w, w4.w, acc[pt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < NTAP; i++) wc[i] = wn[i];
        } else {
        xr[0] = *reinterpret_cast<const float4*>(sb + off[0]);
        xr[1] = *reinterpret_cast<const float4*>(sb + off[NPT > 1 ? 1 : 0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            wr[(tap + WD - 1) % WD] = wload(cb * 9 + tap + WD - 1);
            if (tap < 4) fetch(cb + 1 < L.n_cb ? cb + 1 : cb, tap);
#pragma unroll
            for (int pt = 0; pt < NPT; pt++) {
                const int n = tap * NPT + pt, n2 = n + 2;
                if (n2 < 9 * NPT) {
                    const int tap2 = n2 / NPT, pt2 = n2 - tap2 * NPT;
                    xr[n2 % 3] = *reinterpret_cast<const float4*>(sb + off[pt2] + ((tap2 / 3) * siw + (tap2 % 3)) * 4);
                }
                const float4 x4 = xr[n % 3], w4 = wr[tap % WD];
                acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.x, w4.x, acc[pt], 0, 0, 0);
                acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.y, w4.y, acc[pt], 0, 0, 0);
                acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.z, w4.z, acc[pt], 0, 0, 0);
                acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.w, w4.w, acc[pt], 0, 0, 0);
            }
            // schedule: weight load first, then per pixel tile one LDS read, (one staging load), its four MFMAs
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
            for (int pt = 0; pt < NPT; pt++) {
                if (tap * NPT + pt + 2 < 9 * NPT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (tap < 4 && pt < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // (IN_BNBWD: two loads per tap; otherwise the second group comes up short)
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        (void)FL;
        if (cb + 1 < L.n_cb) transform_store(cb + 1, (cb + 1) & 1);
        __syncthreads();
    }
    // ---- epilogue: lane (q, j) holds pixel slots pt * 16 + 4 q + r (r = 0..3) of output channel 16 cot + j ----
    if (cot >= L.co_tiles) return;  // wave-uniform (no barrier below)
    const int co = cot * 16 + j;
    const bool co_ok = co < L.cout;
    const __amdgpu_buffer_rsrc_t rs_out = mkrs(L.out), rs_skip = mkrs(L.skip ? L.skip : L.out), rs_mask = mkrs(L.mask ? L.mask : L.out),
                                 rs_part = mkrs(L.partner ? L.partner : L.out);
    // (out_plane: G == 1, the workgroup's image is tile (tyi, txi) of plane image pb; offsets stay below 4 GiB: the launcher checks the plane tensor's size)
    constexpr bool oplane = PLANE;
    const int pl_nt = oplane ? L.pl_nty * L.pl_ntx : 1;
    const int pimg = oplane ? img0 / pl_nt : 0, ptl = oplane ? img0 - pimg * pl_nt : 0, tyi = oplane ? ptl / L.pl_ntx : 0, txi = oplane ? ptl - tyi * L.pl_ntx : 0;
    const int pl_hw = L.pl_h * L.pl_w;
    const unsigned soff = oplane ? (unsigned)(((size_t)pimg * L.cout * pl_hw + (size_t)(tyi * L.h) * L.pl_w + (size_t)txi * L.w_img) * sizeof(float))
                                 : (unsigned)((size_t)img0 * L.cout * hw * sizeof(float));
    float ma = 1.0f, mb = 0.0f;
    if (L.mask && L.mcoef && co_ok) { ma = L.mcoef[co]; mb = L.mcoef[L.cpad_out + co]; }
    const bool part_is_mask = L.partner == L.mask;
    float s1 = 0.0f, s2 = 0.0f;
    // ST_FWD: the batch statistics are summed around a PIVOT -- the workgroup's first output of the channel (pixel 0 of its first image, lane
    // group 0) -- so that sum (y - p)^2 carries the variance, not mean^2 (E[y^2] - mean^2 from float32 partial sums loses log10(mean^2 / var) of
    // its seven digits).  The finalize kernel gets (p, sum (y - p), sum (y - p)^2, n) per workgroup and channel.
    const bool st_fwd = L.stat_mode == ST_FWD;
    const float pvt = st_fwd ? __shfl(acc[0][0], j) : 0.0f;
    const int nimg = (img0 + L.G <= L.B) ? L.G : (L.B - img0 > 0 ? L.B - img0 : 0);
    if (st_fwd) {  // (a forward conv carries no skip and no mask: its outputs ARE the accumulators; slot p is a real output iff p < nimg * hw)
        const int nvalid = nimg * hw;  // (workgroup-uniform; the padding channels' sums are never read: the finalize kernel zeroes their coefficients)
#pragma unroll
        for (int pt = 0; pt < NPT; pt++) {
            if ((pt + 1) * 16 <= nvalid) {  // a whole tile of real outputs (all but the last one or two): no per-element test
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float u = acc[pt][r] - pvt;
                    s1 += u;
                    s2 = fmaf(u, u, s2);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float u = (pt * 16 + 4 * q + r) < nvalid ? acc[pt][r] - pvt : 0.0f;
                    s1 += u;
                    s2 = fmaf(u, u, s2);
                }
            }
        }
    }
    constexpr int EC = NPT < 3 ? NPT : 3;
    constexpr int NB = (NPT + EC - 1) / EC;
    // Round 6: where a batch of EC tiles is REGULAR -- every lane's four slots are one whole quad of one image, all real or all padding: the image's
    // pixel count is a multiple of four, or (one image per workgroup) the tiles lie inside it -- its skip / mask / partner operands are loaded with
    // unconditional 16-byte loads (padding lanes read a clamped address and discard), and the loads of batch it + 1 are issued BEFORE batch it is
    // finished and stored: the epilogue of a data-gradient conv was five dependent round trips to L2 / HBM (138.9 us against the forward conv's 130.6 on
    // the 15 x 15 net).  nfast: the leading regular batches (workgroup-uniform; a compile-time fact in the SIDE builds).
    const bool quads_whole = (hw & 3) == 0;
    int nfast = 0;
#pragma unroll
    for (int it = 0; it < NB; it++)
        if (nfast == it && (quads_whole || (L.G == 1 && (it * EC + EC) * 16 <= hw))) nfast = it + 1;  // (the tile path's tiles are 12 or 16 wide: whole quads)
    if (MZLC_EPI_PIPE == 0) nfast = 0;
    struct Bt { f32x4 kv[EC], mv[EC], yv[EC]; unsigned vo[EC]; bool valid[EC]; };
    auto issue = [&](int pb, Bt& b) {
#pragma unroll
        for (int e = 0; e < EC; e++) {
            const int p = (pb + e) * 16 + 4 * q;
            const int g = lc_idiv(p, r_hw), pp = p - g * hw;
            b.valid[e] = co_ok && (pb + e < NPT) && (g < L.G) && (img0 + g < L.B);
            const int gs = b.valid[e] ? g : 0, cs = co_ok ? co : 0, ps = b.valid[e] ? pp : 0;  // (a real address for every lane)
            if (oplane) {
                const int py = lc_idiv(ps, r_iw), px = ps - py * L.w_img;
                b.vo[e] = (unsigned)(cs * pl_hw + py * L.pl_w + px) * (unsigned)sizeof(float);
            } else
            b.vo[e] = (unsigned)((gs * L.cout + cs) * hw + ps) * (unsigned)sizeof(float);
            b.kv[e] = b.mv[e] = b.yv[e] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (pb + e < NPT) {
                if (L.skip) { const float4 t = ld4(rs_skip, b.vo[e], soff); b.kv[e] = f32x4{t.x, t.y, t.z, t.w}; }
                if (L.mask) { const float4 t = ld4(rs_mask, b.vo[e], soff); b.mv[e] = f32x4{t.x, t.y, t.z, t.w}; }
                if (L.partner && !part_is_mask) { const float4 t = ld4(rs_part, b.vo[e], soff); b.yv[e] = f32x4{t.x, t.y, t.z, t.w}; }
            }
        }
    };
    auto finish = [&](int pb, const Bt& b) {
#pragma unroll
        for (int e = 0; e < EC; e++) {
            if (pb + e >= NPT) continue;
            f32x4 v = acc[pb + e];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float t = v[r] + (b.valid[e] ? b.kv[e][r] : 0.0f);
                const float mvr = b.valid[e] ? b.mv[e][r] : 0.0f;
                if (L.mask && !(fmaf(ma, mvr, mb) > 0.0f)) t = 0.0f;
                if (!b.valid[e]) t = 0.0f;
                const float pv = b.valid[e] ? (part_is_mask ? b.mv[e][r] : b.yv[e][r]) : 0.0f;
                if (L.stat_mode == ST_BWD) {
                    s1 += t;
                    s2 = fmaf(t, pv, s2);
                }
                v[r] = t;
            }
            if (b.valid[e])
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rs_out,
                                                       b.vo[e], soff, 0);
        }
    };
    Bt bt[2];
    if (nfast > 0) issue(0, bt[0]);
#pragma unroll
    for (int it = 0; it < NB; it++) {
        const int pb = it * EC;
        if (it < nfast) {
            if (it + 1 < nfast) issue(pb + EC, bt[(it + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            finish(pb, bt[it & 1]);
            __builtin_amdgcn_sched_barrier(0);
            continue;
        }
        unsigned vo[EC][4];
        bool ok[EC][4], vec[EC];
        f32x4 kv[EC], mv[EC], yv[EC];
#pragma unroll
        for (int e = 0; e < EC; e++) {
            const int pt = pb + e < NPT ? pb + e : NPT - 1;
            const int p = pt * 16 + 4 * q;
            int g = lc_idiv(p, r_hw), pp = p - g * hw;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                ok[e][r] = co_ok && (pb + e < NPT) && (g < L.G) && (img0 + g < L.B);
                if (oplane) {
                    const int py = lc_idiv(pp, r_iw), px = pp - py * L.w_img;
                    vo[e][r] = (unsigned)(co * pl_hw + py * L.pl_w + px) * (unsigned)sizeof(float);
                } else
                vo[e][r] = (unsigned)((g * L.cout + co) * hw + pp) * (unsigned)sizeof(float);
                pp++;
                if (pp == hw) { pp = 0; g++; }
            }
            vec[e] = ok[e][0] && ok[e][1] && ok[e][2] && ok[e][3] && vo[e][3] == vo[e][0] + 12;
            kv[e] = mv[e] = yv[e] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto ldv = [&](const __amdgpu_buffer_rsrc_t rs, f32x4& d) {
                if (vec[e]) {
                    const float4 t = ld4(rs, vo[e][0], soff);
                    d = f32x4{t.x, t.y, t.z, t.w};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (ok[e][r]) d[r] = ld1(rs, vo[e][r], soff);
                }
            };
            if (L.skip) ldv(rs_skip, kv[e]);
            if (L.mask) ldv(rs_mask, mv[e]);
            if (L.partner && !part_is_mask) ldv(rs_part, yv[e]);
        }
#pragma unroll
        for (int e = 0; e < EC; e++) {
            if (pb + e >= NPT) continue;
            f32x4 v = acc[pb + e];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float t = v[r] + kv[e][r];
                if (L.mask && !(fmaf(ma, mv[e][r], mb) > 0.0f)) t = 0.0f;
                if (!ok[e][r]) t = 0.0f;
                const float pv = part_is_mask ? mv[e][r] : yv[e][r];
                if (L.stat_mode == ST_BWD) {
                    s1 += t;
                    s2 = fmaf(t, pv, s2);
                }
                v[r] = t;
            }
            if (vec[e]) {
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rs_out,
                                                       vo[e][0], soff, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (ok[e][r]) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), rs_out, vo[e][r], soff, 0);
            }
        }
    }
    if (L.stat_mode != ST_NONE) {  // the four lane groups q hold disjoint pixels of channel j: (q0 + q1) + (q2 + q3)
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (q == 0 && co < L.cpad_out) {
            if (st_fwd) {
                *reinterpret_cast<float4*>(L.stat_part + ((size_t)by * L.cpad_out + co) * 4) = make_float4(co_ok ? pvt : 0.0f, s1, s2, co_ok ? (float)(nimg * hw) : 0.0f);
            } else {
                float* d = L.stat_part + ((size_t)by * L.cpad_out + co) * 2;
                d[0] = s1; d[1] = s2;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradient.  workgroup = 4 waves = a 32 x 32 block (co x ci) of one conv layer x all 9 taps x one chunk of images; wave (wm, wn) owns
// the 16 x 16 tile (co tile 2 cob + wm, ci tile 2 cib + wn) for the 9 taps: 9 accumulators.  The reduction index of the MFMAs is the PIXEL:
// both operands are staged per image into LDS planes [channel][flat position] with row pitch P4 = 4 ceil((w + 1) / 4) and zero padding (at
// least one zero column behind every row, zero rows above and below), so that the flat position f + dy P4 + dx IS the tap's neighbour and
// a pad position contributes 0.  Lane (i, kq) multiplies positions 16 g + 4 kq + s in k-step s: dy is one aligned 16-byte read, and the tap
// shifts dx = -1 / +1 of x are the same quad with one element taken from the neighbouring dword -- a register choice.
// ---------------------------------------------------------------------------------------------------------------------------------
struct LcWgradSrc {        // one unroll step's operands of a layer the steps share (see LcWgrad::srcs)
    const float *dz, *y, *dcoef, *x0, *xcoef;
};
struct LcWgrad {
    const float* dz;       // [B][cout][hw]
    const float* y;        // [B][cout][hw]
    const float* dcoef;    // [3][cpad_out]: dy = c1 dz + c2 y + c3
    const float* x0;       // [B][cin_real][hw]
    const float* xcoef;    // [3][cpad_in] (x_mode == IN_BNRELU: a, b)
    const int* action;     // [B] or null
    float* part;           // [chunks][9][co_pad][ci_pad]
    int x_mode, num_actions;
    int cin_real, cin, cout, ci_tiles, co_tiles, cpad_in, cpad_out;
    int B, ipw;            // images per chunk
    int h, w_img, P4, nsteps, SPY, SPX;
    int co_blocks;         // blockIdx.y = chunk * co_blocks + cob
    int sg;                // images per staging round (nsteps / SPY / SPX / P4 are the round's)
    int sg_cols;           // 1: the round's images side by side (column offset gi (w + 1)); 0: stacked vertically (row offset gi (h + 1))
    int tapmask;           // 0 / 0x1ff: all nine taps; else only these accumulators exist (a parity plane of a stride-2 conv; see k_lc_conv's TAPMASK)
    const LcWgradSrc* srcs;  // round 6, or null: the layer's operands of `nsrc` unroll steps (device table) -- the dynamics / prediction towers apply the same
    int nsrc, cps;           // weights at every step, and on a 6 x 6 plane one step's batch is two staging rounds per workgroup: prologue, write-out and the
                             // reduce launch cost more than the MFMAs.  All steps in ONE launch: chunk c is chunk c % cps of step c / cps (B images per step;
                             // dz / y / dcoef / x0 / xcoef above are not read), `cps` chunks per step.
    int ring_zero;         // 1: the outermost ring of every dy image is multiplied by 0 (tiles gathered with their halo: only the inner pixels are outputs)
    int ring_rows;         // round 6, with ring_zero: 1: the dy plane holds rows 1 .. h - 2 only -- the first and last row of a haloed tile are all ring, i.e. all
                           // zeros, and were 2 P4 of the reduction's positions (18 -> 15 steps of 16 positions for a 14 x 18 tile, 14 -> 12 for 14 x 14) --
                           // and the x plane's "zero row above / below" slots carry the tile's real rows 0 and h - 1.  nsteps / SPY / SPX are those of h - 2 rows.
                           // 2: and a reduction step is one ROW of the tile's inner columns (w - 2 <= 16 of them) instead of 16 consecutive flat positions:
                           // both planes are stored one position to the left, so that the inner columns 1 .. w - 2 are slots 0 .. of their pitch row and
                           // step g reads the aligned quads of row g; the ring columns' dy (zeros) have no slot, the pad slots are never multiplied:
                           // nsteps = h - 2 (a 14 x 18 tile, pitch 20: 12 steps for 15 -- a fifth of the MFMAs were pad and ring columns).
                           // 3: as 2 for an inner width of 12 (pitch 16), and a step is the next four REAL quads in row order (see `request`): nsteps = 3 (h - 2) / 4.
};

// ACT: the layer has action-plane input channels (the dynamics net's first conv); the other builds carry none of that code.
// LDS is single-buffered (67.6 KB at 15 x 15): TWO workgroups share a CU, one's staging (global -> registers is in flight during the MFMAs, but the
// transform and the LDS writes are VALU / LDS issue) runs under the other's MFMAs.
template <bool ACT, bool RING = false, int TAPMASK = 0x1ff>
__global__ __launch_bounds__(256, 2) void k_lc_wgrad(const Pair<LcWgrad> PJ) {
    // XCD-aware placement.  The 8 XCDs take workgroups round-robin by linear id, each with its own 4 MB L2; the co_blocks x ci_blocks workgroups
    // of one image chunk read the SAME dz / y / x planes (each plane is staged by every block of the other channel dimension).  In launch order
    // they land on eight different XCDs and every one fetches its own copy from the MALL: 270 MB per paired launch against 88 MB algorithmic
    // (PMC, profiles/round5/convlearner).  Remapped, a chunk's blocks share an XCD: id L -> xcd = L & 7, group = xcd + 8 * ((L >> 3) / blocks).
    int bx = blockIdx.x, byy = blockIdx.y;
    if (PJ.remap) {
        const int X = gridDim.x, Yc = PJ.a.co_blocks, nblocks = X * Yc;
        const int Lid = (int)blockIdx.x + X * (int)blockIdx.y, s = Lid >> 3;
        const int g = (Lid & 7) + 8 * (s / nblocks), blk = s % nblocks;
        bx = blk % X;
        byy = g * Yc + blk / X;
    }
    const bool second = byy >= PJ.na;
    LcWgrad L = second ? PJ.b : PJ.a;
    const int by = second ? byy - PJ.na : byy;
    int lchunk = by / L.co_blocks;  // the chunk within its unroll step
    if (L.srcs) {
        const int s = lchunk / L.cps;
        lchunk -= s * L.cps;
        const LcWgradSrc S = L.srcs[s];
        L.dz = S.dz; L.y = S.y; L.dcoef = S.dcoef; L.x0 = S.x0; L.xcoef = S.xcoef;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sm = reinterpret_cast<float*>(smem);
    const int ybuf = 32 * L.SPY, xbuf = 32 * L.SPX;
    float* s_y = sm;                // [32][SPY]: position f of the padded image at f - P4 (rows 0 .. h - 1 only)
    float* s_x = sm + ybuf;         // [32][SPX]: position f at f + 4 (4 floats of margin in front)
    float* s_dc = s_x + xbuf;       // [3][32] dy coefficients of this block's channels
    float* s_xc = s_dc + 96;        // [2][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kq = lane >> 4, i16 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    const int cib = bx, cob = by % L.co_blocks, chunk = by / L.co_blocks;
    if (cib * 2 >= L.ci_tiles) return;  // (a paired job with fewer input-channel blocks; workgroup-uniform, before any barrier)
    const int hw = L.h * L.w_img, QP = (hw + 3) >> 2;
    const float r_iw = 1.0f / (float)L.w_img;
    const int co0 = cob * 32, ci0 = cib * 32;
    for (int i = tid; i < (ybuf + xbuf) / 4; i += 256) reinterpret_cast<float4*>(sm)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 96) {
        const int c = co0 + (tid & 31);
        s_dc[tid] = c < L.cout ? L.dcoef[(tid >> 5) * L.cpad_out + c] : 0.0f;  // (channels past cout: dy = 0 * dz + 0 * y + 0)
    }
    if (tid >= 128 && tid < 192) {
        const int t = tid - 128, c = ci0 + (t & 31);
        s_xc[t] = (L.x_mode == IN_BNRELU && c < L.cin_real) ? L.xcoef[(t >> 5) * L.cpad_in + c] : 0.0f;
    }
    const __amdgpu_buffer_rsrc_t rs_dz = mkrs(L.dz), rs_y = mkrs(L.y), rs_x = mkrs(L.x0);
    // staging plan: lane = (image of the round, pixel quad) (lanes >= sg QP idle), wave w the channels w, 4 + w, .., 28 + w of each operand.
    // Elements past the image (the last quad) get their own exec-mask region per element index: 8 lane branches per round instead of one per store
    const int gi = lc_idiv(lane, 1.0f / (float)QP), ql = lane - gi * QP;
    const bool s_ok = gi < L.sg;
    const int p0 = (s_ok ? ql : 0) * 4;
    int spos[4], pm[4];
    float ym[4];
    bool sx_ok[4];  // this element is staged
    int sposy[4];   // its position in the dy plane (ring_rows: the plane's spare tail for the rows that have none)
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int pp = p0 + e, py = lc_idiv(pp, r_iw), px = pp - py * L.w_img;
        spos[e] = (s_ok && pp < hw) ? (L.sg_cols ? py * L.P4 + gi * (L.w_img + 1) + px : (gi * (L.h + 1) + py) * L.P4 + px) : -1;  // + P4 + 4 in the x planes (one zero row above, the margin)
        pm[e] = ACT ? pp % L.num_actions : 0;
        ym[e] = (RING && (py == 0 || px == 0 || py == L.h - 1 || px == L.w_img - 1)) ? 0.0f : 1.0f;
        sx_ok[e] = spos[e] >= 0;
        sposy[e] = spos[e];
        if (RING && L.ring_rows) {  // (one image per round: the launcher checks) row py of the tile is row py - 1 of the planes; rows 0 and h - 1 exist in x only
            const bool rows2 = L.ring_rows >= 2;
            spos[e] -= rows2 ? L.P4 + 1 : L.P4;
            sposy[e] = (py >= 1 && py <= L.h - 2 && (!rows2 || (px >= 1 && px <= L.w_img - 2))) ? spos[e] : L.SPY - 4;
        }
    }
    float4 rdz[8], ry[8], rx[8];
    int r_act = -1;
    bool r_ok = false;  // this lane's image of the round in flight belongs to the chunk (the last round may be short: its idle slots are staged as zeros)
    const int b_lo = lchunk * L.ipw, b_hi = (b_lo + L.ipw < L.B) ? b_lo + L.ipw : L.B;
    auto fetch = [&](int b) {  // round of images b .. b + sg - 1
        const int bi = b + (s_ok ? gi : 0);
        r_ok = s_ok && bi < b_hi;
        const int rel = r_ok ? bi - b : 0;
        const unsigned vo_o = (unsigned)((rel * L.cout * hw + p0) * sizeof(float)), vo_i = (unsigned)((rel * L.cin_real * hw + p0) * sizeof(float));
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = co0 + 4 * k + wave, cc = c < L.cout ? c : 0;
            const int so = (int)(((size_t)b * L.cout + cc) * hw * sizeof(float));
            rdz[k] = ld4(rs_dz, vo_o, so);
            ry[k] = ld4(rs_y, vo_o, so);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = ci0 + 4 * k + wave, cc = c < L.cin_real ? c : 0;
            rx[k] = ld4(rs_x, vo_i, (int)(((size_t)b * L.cin_real + cc) * hw * sizeof(float)));
        }
        if (ACT) r_act = L.action[b + rel];
    };
    auto stage = [&]() {
        float vy[8][4], vx[8][4];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int cl = 4 * k + wave;
            const float c1 = s_dc[cl], c2 = s_dc[32 + cl], c3 = s_dc[64 + cl];
            const float a[4] = {rdz[k].x, rdz[k].y, rdz[k].z, rdz[k].w}, yy[4] = {ry[k].x, ry[k].y, ry[k].z, ry[k].w};
#pragma unroll
            for (int e = 0; e < 4; e++) vy[k][e] = RING ? fmaf(c1, a[e], fmaf(c2, yy[e], c3)) * ym[e] : fmaf(c1, a[e], fmaf(c2, yy[e], c3));
            if (!r_ok) {  // (an idle image slot of a short last round: zeros over what the previous round left there)
#pragma unroll
                for (int e = 0; e < 4; e++) vy[k][e] = 0.0f;
            }
        }
        const bool bnrelu = L.x_mode == IN_BNRELU;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int cl = 4 * k + wave, c = ci0 + cl;
            const float x[4] = {rx[k].x, rx[k].y, rx[k].z, rx[k].w};
            if (!ACT || c < L.cin_real) {
                if (c >= L.cin_real) {  // (padding channels of a layer without action planes; wave-uniform)
#pragma unroll
                    for (int e = 0; e < 4; e++) vx[k][e] = 0.0f;
                } else if (bnrelu) {
                    const float xa = s_xc[cl], xb = s_xc[32 + cl];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float t = fmaf(xa, x[e], xb);
                        vx[k][e] = t > 0.0f ? t : 0.0f;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) vx[k][e] = x[e];
                }
            } else {  // action planes (network.py:440-444)
                const int t = c < L.cin ? (int)(((long long)(c - L.cin_real) * hw) % L.num_actions) : 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    int m = pm[e] + t;
                    m = m >= L.num_actions ? m - L.num_actions : m;
                    vx[k][e] = (c < L.cin && m == r_act) ? 1.0f : 0.0f;
                }
            }
        }
        float* dx_ = s_x + L.P4 + 4;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            // (the loop of round 5, unconditional stores: a per-store predicate here cost the board net's update 6 % when it was tried.  ring_rows: the
            // tile's rows 0 and h - 1 have no dy slot -- their dy values, zeros, go to the plane's unused 4-float tail at SPY - 4)
            if (sx_ok[e]) {
                float* py = s_y + sposy[e] + wave * L.SPY;
                float* px = dx_ + spos[e] + wave * L.SPX;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    py[4 * k * L.SPY] = vy[k][e];
                    px[4 * k * L.SPX] = vx[k][e];
                }
            }
        }
    };
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (b_lo < b_hi) fetch(b_lo);
    const int SGn = L.sg;
    const float* py_ = s_y + (wm * 16 + i16) * L.SPY + 4 * kq;             // + 16 g: the lane's dy quad
    const float* px_ = s_x + (wn * 16 + i16) * L.SPX + L.P4 + 4 + 4 * kq;   // + 16 g + dy P4: the centre quad of row dy
    const int P4 = L.P4;
    const int GS = (RING && L.ring_rows >= 2) ? P4 : 16;  // floats from one reduction step's quads to the next one's
    for (int b = b_lo; b < b_hi; b += SGn) {
        __syncthreads();  // the previous round's MFMAs have read the planes (first pass: the zero fill is complete)
        stage();
        __syncthreads();
        if (b + SGn < b_hi) fetch(b + SGn);  // in flight during this round's MFMAs
        // operands of step g + 1 are requested before the MFMAs of step g: two register sets, the loop unrolled by two, scheduling barriers between
        // "request" and "multiply" (left alone the scheduler sinks the reads to their first use and every step waits for its own LDS round trip)
        struct OpSet { float4 a4; float4 c4[3]; float lf[3], rg[3]; };
        auto request = [&](int g, OpSet& o) {
            const int gc = g < L.nsteps ? g : L.nsteps - 1;
            // (ring_rows == 3: the tile's inner width is 12 -- three quads a row at pitch 16 -- and a step takes the next FOUR REAL quads, row by row: quad
            // Q = 4 g + kq of the 3 (h - 2) is row Q / 3, columns 4 (Q % 3); the pad quad of every row is never multiplied: 9 steps for 12 at 14 x 14)
            int og = GS * gc;
            if (RING && L.ring_rows == 3) {
                const int Q = 4 * gc + kq, row = (Q * 43691) >> 17;
                og = row * P4 + 4 * (Q - 3 * row) - 4 * kq;  // (py_ / px_ carry + 4 kq)
            }
            o.a4 = *reinterpret_cast<const float4*>(py_ + og);
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                if (((TAPMASK >> (3 * dy)) & 7) != 0) {  // (a row none of whose taps exists is not read)
                    const float* r = px_ + og + (dy - 1) * P4;
                    o.c4[dy] = *reinterpret_cast<const float4*>(r);
                    if ((TAPMASK >> (3 * dy)) & 1) o.lf[dy] = r[-1];
                    if ((TAPMASK >> (3 * dy + 2)) & 1) o.rg[dy] = r[4];
                }
            }
        };
        auto multiply = [&](const OpSet& o) {
            const float a[4] = {o.a4.x, o.a4.y, o.a4.z, o.a4.w};
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                float xs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (((TAPMASK >> (3 * dy)) & 7) != 0) {
                    xs[1] = o.c4[dy].x; xs[2] = o.c4[dy].y; xs[3] = o.c4[dy].z; xs[4] = o.c4[dy].w;
                    if ((TAPMASK >> (3 * dy)) & 1) xs[0] = o.lf[dy];
                    if ((TAPMASK >> (3 * dy + 2)) & 1) xs[5] = o.rg[dy];
                }
#pragma unroll
                for (int dx = 0; dx < 3; dx++) {
                    if ((TAPMASK >> (dy * 3 + dx)) & 1) {
#pragma unroll
                        for (int s = 0; s < 4; s++) acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], xs[s + dx], acc[dy * 3 + dx], 0, 0, 0);
                    }
                }
            }
        };
        OpSet o0, o1;
        request(0, o0);
        for (int g = 0; g < L.nsteps; g += 2) {
            request(g + 1, o1);
            __builtin_amdgcn_sched_barrier(0);
            multiply(o0);
            __builtin_amdgcn_sched_barrier(0);
            request(g + 2, o0);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < L.nsteps) multiply(o1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D[m = co][n = ci]: lane (kq, i16) holds rows 4 kq + r of column i16
    const int cot = cob * 2 + wm, cit = cib * 2 + wn;
    if (cot >= L.co_tiles || cit >= L.ci_tiles) return;
    const int co_pad = L.co_tiles * 16, ci_pad = L.ci_tiles * 16;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        if ((TAPMASK >> t) & 1) {  // (the other tap planes of the partials are never read as values: k_lc_wreduce's tap map drops them)
            float* d = L.part + (((size_t)chunk * 9 + t) * co_pad + cot * 16 + 4 * kq) * ci_pad + cit * 16 + i16;
#pragma unroll
            for (int r = 0; r < 4; r++) d[(size_t)r * ci_pad] = acc[t][r];
        }
    }
}

// chunk partials -> gradient of the conv weight in torch layout [cout][cin][3][3]; accumulate: += (later unroll steps of a shared layer)
struct LcWreduce {
    const float* part;
    float* grad;
    int chunks, cout, cin, co_pad, ci_pad, accumulate;
    int cin_loop;           // input channels the partials cover (0: all `cin`; the dynamics net's first conv: its hidden channels -- k_lc_wgrad_act does the action planes)
    int use_map;            // 1: accumulator tap t of the partials is weight tap tapmap[t] (-1: not a tap of this parity plane; see LcTileGather)
    signed char tapmap[9];
};
// thread = one (co, ci) pair, all nine taps: the partial planes are read coalesced over ci (four chunks x nine taps in flight), the nine
// results are 36 contiguous bytes of the gradient (one thread per (tap, co, ci) wrote them 36 bytes apart).  Chunks are added in order.
__global__ __launch_bounds__(64) void k_lc_wreduce(const Pair<LcWreduce> PJ) {
    const bool second = (int)blockIdx.y >= PJ.na;
    const LcWreduce L = second ? PJ.b : PJ.a;
    const int by = second ? (int)blockIdx.y - PJ.na : (int)blockIdx.y;
    const int i = by * 64 + threadIdx.x;  // (co, ci), ci fastest
    const int cl = L.cin_loop ? L.cin_loop : L.cin;
    if (i >= L.cout * cl) return;
    const int ci = i % cl, co = i / cl;
    const size_t ts = (size_t)L.co_pad * L.ci_pad, cs = 9 * ts;
    const float* p = L.part + (size_t)co * L.ci_pad + ci;
    float s[9];
#pragma unroll
    for (int t = 0; t < 9; t++) s[t] = 0.0f;
    for (int c = 0; c < L.chunks; c += 4) {
        float v[4][9];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int cc = c + k < L.chunks ? c + k : L.chunks - 1;
#pragma unroll
            for (int t = 0; t < 9; t++) v[k][t] = p[cc * cs + t * ts];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (c + k < L.chunks) {
#pragma unroll
                for (int t = 0; t < 9; t++) s[t] += v[k][t];
            }
        }
    }
    float* g = L.grad + ((size_t)co * L.cin + ci) * 9;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int tap = L.use_map ? (int)L.tapmap[t] : t;
        if (tap >= 0) g[tap] = L.accumulate ? g[tap] + s[t] : s[t];
    }
}

// Weight gradient of the ACTION planes of the dynamics net's first conv (network.py:440-444).  Plane c' of image b is one where
// (p + c' hw) mod A == action_b and zero elsewhere -- hw / A positions (at most one for the board games, A = hw + 1) -- so
//     dW[co][cin_real + c'][ky][kx] = sum_b sum_{those q} dy[b][co][q - (ky - 1, kx - 1)]        (inside the image)
// is a gather of dy, not a matrix product: as MFMA work the 226 planes of the C5 net were 65 % of that layer's weight gradient.
// grid (cout, batch chunks), thread = plane c'; consecutive planes read consecutive positions.  Partials per chunk, reduced in chunk order.
struct LcWgradAct {
    const float* dz;       // [B][cout][hw]
    const float* y;        // [B][cout][hw]
    const float* dcoef;    // [3][cpad_out]: dy = c1 dz + c2 y + c3
    const int* action;     // [B]
    float* part;           // [nchunk][cout][A][9]
    float* grad;           // the layer's weight gradient [cout][cin][9]
    int B, cout, cpad_out, cin, cin_real, A, h, w, bchunk, nchunk, accumulate;
};
__global__ __launch_bounds__(256) void k_lc_wgrad_act(const LcWgradAct L) {
    const int co = blockIdx.x, ck = blockIdx.y, cp = threadIdx.x;
    if (cp >= L.A) return;
    const float c1 = L.dcoef[co], c2 = L.dcoef[L.cpad_out + co], c3 = L.dcoef[2 * L.cpad_out + co];
    const int hw = L.h * L.w, shift = (int)(((long long)cp * hw) % L.A);
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++) acc[t] = 0.0f;
    const int b_lo = ck * L.bchunk, b_hi = b_lo + L.bchunk < L.B ? b_lo + L.bchunk : L.B;
    for (int b = b_lo; b < b_hi; b++) {
        int p0 = L.action[b] - shift;
        p0 = p0 < 0 ? p0 + L.A : p0;
        const float* dzb = L.dz + ((size_t)b * L.cout + co) * hw;
        const float* yb = L.y + ((size_t)b * L.cout + co) * hw;
        for (int q = p0; q < hw; q += L.A) {
            const int yq = q / L.w, xq = q - yq * L.w;
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    const int py = yq - ky + 1, px = xq - kx + 1;
                    if (py >= 0 && py < L.h && px >= 0 && px < L.w) {
                        const int pp = py * L.w + px;
                        acc[ky * 3 + kx] += fmaf(c1, dzb[pp], fmaf(c2, yb[pp], c3));
                    }
                }
            }
        }
    }
    float* d = L.part + (((size_t)ck * L.cout + co) * L.A + cp) * 9;
#pragma unroll
    for (int t = 0; t < 9; t++) d[t] = acc[t];
}
__global__ __launch_bounds__(256) void k_lc_wgrad_act_reduce(const LcWgradAct L) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // (co, c', tap)
    if (i >= L.cout * L.A * 9) return;
    const int t = i % 9, cp = (i / 9) % L.A, co = i / (9 * L.A);
    float s = 0.0f;
    for (int ck = 0; ck < L.nchunk; ck++) s += L.part[(size_t)ck * L.cout * L.A * 9 + i];
    float* g = L.grad + ((size_t)co * L.cin + L.cin_real + cp) * 9 + t;
    *g = L.accumulate ? *g + s : s;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// BatchNorm2d in train mode (network.py:283-291): finalize the per-workgroup partial sums.  One thread per channel; the groups are added
// in order in float64 (the partials are float32 sums of <= a few hundred values).
// ---------------------------------------------------------------------------------------------------------------------------------
struct LcBnFwd {
    const float* part;     // [groups][cpad][4]: pivot p, sum (y - p), sum (y - p)^2, n (k_lc_conv ST_FWD, k_lc_tile_scatter)
    const float* gamma;    // [C]
    const float* beta;
    float* coef;           // [3][cpad]: a, b, (unused)    -- what the next conv's staging applies
    float* save;           // [2][cpad]: mean, invstd       -- for the backward pass
    float* running_mean;   // [C] or null
    float* running_var;
    int64_t* num_batches;  // or null
    int groups, C, cpad;
    float count;           // B * hw
};
// workgroup = 16 channels x 16 group slices: slice j adds groups j, j + 16, .. in float64, the 16 slices meet in LDS in slice order
__device__ __forceinline__ void lc_group_sums(const float* part, int groups, int cpad, int c, int slice, double (*s_acc)[16][2], double& s1, double& s2) {
    double a = 0.0, b = 0.0;
    int g = slice;
    for (; g + 7 * 16 < groups; g += 8 * 16) {  // (the tiled Atari stages have thousands of groups: eight loads in flight, added in group order)
        float2 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = *reinterpret_cast<const float2*>(part + ((size_t)(g + 16 * k) * cpad + c) * 2);
#pragma unroll
        for (int k = 0; k < 8; k++) { a += (double)v[k].x; b += (double)v[k].y; }
    }
    for (; g < groups; g += 16) {
        const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)g * cpad + c) * 2);
        a += (double)v.x;
        b += (double)v.y;
    }
    s_acc[threadIdx.x & 15][slice][0] = a;
    s_acc[threadIdx.x & 15][slice][1] = b;
    __syncthreads();
    s1 = 0.0; s2 = 0.0;
    for (int k = 0; k < 16; k++) { s1 += s_acc[threadIdx.x & 15][k][0]; s2 += s_acc[threadIdx.x & 15][k][1]; }
}

// forward statistics from the pivoted partials: with T1 = sum S1, T2 = sum S2, U1 = sum n p, U2 = sum n p^2, V = sum p S1 over the groups,
//   N mean = U1 + T1,   N var = T2 + 2 V - 2 mean T1 + U2 - 2 mean U1 + mean^2 N     (sum (y - mean)^2 with y - mean = (y - p) + (p - mean))
// every sum in float64, slices and groups in a fixed order
__device__ __forceinline__ void lc_pivot_sums(const float* part, int groups, int cpad, int c, int slice, double (*s_acc)[16][6], double& nmean, double& nvar,
                                              double& ntot) {
    double T0 = 0.0, T1 = 0.0, T2 = 0.0, U1 = 0.0, U2 = 0.0, V = 0.0;
    int g = slice;
    for (; g + 3 * 16 < groups; g += 4 * 16) {  // four loads in flight, added in group order
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const float4*>(part + ((size_t)(g + 16 * k) * cpad + c) * 4);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double p = v[k].x, a = v[k].y, b = v[k].z, n = v[k].w;
            T0 += n; T1 += a; T2 += b; U1 += n * p; U2 += n * p * p; V += p * a;
        }
    }
    for (; g < groups; g += 16) {
        const float4 v = *reinterpret_cast<const float4*>(part + ((size_t)g * cpad + c) * 4);
        const double p = v.x, a = v.y, b = v.z, n = v.w;
        T0 += n; T1 += a; T2 += b; U1 += n * p; U2 += n * p * p; V += p * a;
    }
    double* mine = s_acc[threadIdx.x & 15][slice];
    mine[0] = T0; mine[1] = T1; mine[2] = T2; mine[3] = U1; mine[4] = U2; mine[5] = V;
    __syncthreads();
    T0 = T1 = T2 = U1 = U2 = V = 0.0;
    for (int k = 0; k < 16; k++) {
        const double* o = s_acc[threadIdx.x & 15][k];
        T0 += o[0]; T1 += o[1]; T2 += o[2]; U1 += o[3]; U2 += o[4]; V += o[5];
    }
    ntot = T0;
    nmean = U1 + T1;
    const double mean = T0 > 0.0 ? nmean / T0 : 0.0;
    nvar = T2 + 2.0 * V - 2.0 * mean * T1 + U2 - 2.0 * mean * U1 + mean * mean * T0;
}

__global__ __launch_bounds__(256) void k_lc_bn_fwd(const Pair<LcBnFwd> PJ) {
    __shared__ double s_acc[16][16][6];
    const bool second = (int)blockIdx.y >= PJ.na;
    const LcBnFwd L = second ? PJ.b : PJ.a;
    const int by = second ? (int)blockIdx.y - PJ.na : (int)blockIdx.y;
    const int c = by * 16 + (threadIdx.x & 15), slice = threadIdx.x >> 4;
    const int cc = c < L.cpad ? c : L.cpad - 1;
    double nmean, nvar, ntot;
    lc_pivot_sums(L.part, L.groups, L.cpad, cc, slice, s_acc, nmean, nvar, ntot);
    if (slice != 0 || c >= L.cpad) return;
    if (c >= L.C) {
        L.coef[c] = 0.0f; L.coef[L.cpad + c] = 0.0f; L.coef[2 * L.cpad + c] = 0.0f;
        L.save[c] = 0.0f; L.save[L.cpad + c] = 0.0f;
        return;
    }
    // (ntot == L.count: every valid output position is in exactly one partial)
    const double mean = nmean / (double)L.count;
    double var = nvar / (double)L.count;
    var = var < 0.0 ? 0.0 : var;
    const float invstd = (float)(1.0 / sqrt(var + 1e-5));
    const float a = L.gamma[c] * invstd;
    L.coef[c] = a;
    L.coef[L.cpad + c] = L.beta[c] - (float)mean * a;
    L.coef[2 * L.cpad + c] = 0.0f;
    L.save[c] = (float)mean;
    L.save[L.cpad + c] = invstd;
    if (L.running_mean) {  // momentum 0.1, unbiased variance (torch.nn.BatchNorm2d defaults)
        const double unb = L.count > 1.0f ? var * (double)L.count / ((double)L.count - 1.0) : var;
        L.running_mean[c] = (float)(0.9 * (double)L.running_mean[c] + 0.1 * mean);
        L.running_var[c] = (float)(0.9 * (double)L.running_var[c] + 0.1 * unb);
    }
    if (L.num_batches && c == 0) *L.num_batches += 1;
}

struct LcBnBwd {
    const float* part;     // [groups][cpad][2]: sum dz, sum dz * y
    const float* gamma;
    const float* save;     // mean, invstd
    float* coef;           // [3][cpad]: dy = c1 dz + c2 y + c3
    float* dgamma;         // gradient slots
    float* dbeta;
    int groups, C, cpad, accumulate;
    float count;
};
__global__ __launch_bounds__(256) void k_lc_bn_bwd(const Pair<LcBnBwd> PJ) {
    __shared__ double s_acc[16][16][2];
    const bool second = (int)blockIdx.y >= PJ.na;
    const LcBnBwd L = second ? PJ.b : PJ.a;
    const int by = second ? (int)blockIdx.y - PJ.na : (int)blockIdx.y;
    const int c = by * 16 + (threadIdx.x & 15), slice = threadIdx.x >> 4;
    const int cc = c < L.cpad ? c : L.cpad - 1;
    double s1, s2;
    lc_group_sums(L.part, L.groups, L.cpad, cc, slice, s_acc, s1, s2);
    if (slice != 0 || c >= L.cpad) return;
    if (c >= L.C) {
        L.coef[c] = 0.0f; L.coef[L.cpad + c] = 0.0f; L.coef[2 * L.cpad + c] = 0.0f;
        return;
    }
    const double mean = L.save[c], invstd = L.save[L.cpad + c], gam = L.gamma[c], M = L.count;
    const double dgam = (s2 - mean * s1) * invstd;
    const double c1 = gam * invstd;
    L.coef[c] = (float)c1;
    L.coef[L.cpad + c] = (float)(-c1 * invstd * dgam / M);
    L.coef[2 * L.cpad + c] = (float)(-c1 * s1 / M + c1 * mean * invstd * dgam / M);
    L.dgamma[c] = L.accumulate ? L.dgamma[c] + (float)dgam : (float)dgam;
    L.dbeta[c] = L.accumulate ? L.dbeta[c] + (float)s1 : (float)s1;
}

// x' = relu(a y + b [+ res]) materialised (block outputs, ResNetBlock network.py:293-299; the first conv block's output)
struct LcApply {
    const float* y;
    const float* res;   // or null
    const float* coef;  // [3][cpad]
    float* out;
    int C, hw, cpad;
    long long n;        // B * C * hw
};
__global__ __launch_bounds__(256) void k_lc_apply(const Pair<LcApply> PJ) {
    const bool second = (int)blockIdx.y >= PJ.na;
    const LcApply L = second ? PJ.b : PJ.a;
    const int by = second ? (int)blockIdx.y - PJ.na : (int)blockIdx.y;
    const long long i0 = ((long long)by * 256 + threadIdx.x) * 4;
    if (i0 >= L.n) return;
    const float r_hw = 1.0f / (float)L.hw;
    if (i0 + 3 < L.n) {
        const float* py = L.y + i0;
        float yv[4] = {py[0], py[1], py[2], py[3]}, rv[4] = {0.f, 0.f, 0.f, 0.f};
        if (L.res) { rv[0] = L.res[i0]; rv[1] = L.res[i0 + 1]; rv[2] = L.res[i0 + 2]; rv[3] = L.res[i0 + 3]; }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const long long i = i0 + e;
            const int c = (int)((i / L.hw) % L.C);
            const float t = fmaf(L.coef[c], yv[e], L.coef[L.cpad + c]) + rv[e];
            L.out[i] = t > 0.0f ? t : 0.0f;
        }
        (void)r_hw;
        return;
    }
    for (long long i = i0; i < L.n; i++) {
        const int c = (int)((i / L.hw) % L.C);
        const float t = fmaf(L.coef[c], L.y[i], L.coef[L.cpad + c]) + (L.res ? L.res[i] : 0.0f);
        L.out[i] = t > 0.0f ? t : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// normalize_hidden_state (util.py:31-36) over the channels of each pixel.  workgroup = one image x 32 pixels; thread (pixel, channel group of 8).
// forward: s = (x - min) / (max - min + 1e-8).
// k_lc_entry: the gradient entering a tower's output x (post-ReLU): g = [normalize backward of `gs` scaled by `scale`] + `extra`, masked
// by x > 0, plus the partial sums (sum dz, sum dz y) of the last BatchNorm of the tower.  normalize backward (min / max gradients go to
// the FIRST attaining channel, as torch's min(dim) / max(dim) backward does):
//     dx_c = G_c / d - [c == imin] sum_k G_k / d - ([c == imax] - [c == imin]) sum_k G_k s_k / d,     d = max - min + 1e-8
// ---------------------------------------------------------------------------------------------------------------------------------
template <int CPT>
__global__ __launch_bounds__(256) void k_lc_normalize(const float* in, float* out, int B, int C, int hw) {
    __shared__ float s_mn[8][32], s_mx[8][32];
    const int b = blockIdx.y, px = threadIdx.x & 31, cg = threadIdx.x >> 5, p = blockIdx.x * 32 + px;
    const bool ok = p < hw;
    const int cpt = (C + 7) >> 3;
    const float* s = in + (size_t)b * C * hw + (ok ? p : 0);
    float v[CPT];
    float mn = __uint_as_float(0x7f800000u), mx = __uint_as_float(0xff800000u);
#pragma unroll
    for (int i = 0; i < CPT; i++) {
        const int c = cg * cpt + i;
        if (i < cpt && c < C) {
            v[i] = s[(size_t)c * hw];
            mn = v[i] < mn ? v[i] : mn;
            mx = v[i] > mx ? v[i] : mx;
        }
    }
    s_mn[cg][px] = mn; s_mx[cg][px] = mx;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 8; g++) {
        mn = s_mn[g][px] < mn ? s_mn[g][px] : mn;
        mx = s_mx[g][px] > mx ? s_mx[g][px] : mx;
    }
    if (!ok) return;
    const float d = (mx - mn) + 1e-8f;
    float* o = out + (size_t)b * C * hw + p;
#pragma unroll
    for (int i = 0; i < CPT; i++) {
        const int c = cg * cpt + i;
        if (i < cpt && c < C) o[(size_t)c * hw] = (v[i] - mn) / d;
    }
}

struct LcEntry {
    const float* x;        // tower output (post-ReLU) [B][C][hw]
    const float* gs;       // gradient wrt normalize(x), or null
    const float* extra;    // gradient added directly (head backward), or null
    const float* partner;  // y of the tower's last BatchNorm
    float* dz;             // out
    float* stat_part;      // [B * chunks][cpad][2]
    float scale;           // on gs (0.5: pipeline.py:584)
    int B, C, hw, cpad;
};
// grid (nsplit, B): workgroup (s, b) walks the 32-pixel chunks s, s + nsplit, .. of image b; its per-channel partial sums stay in registers
// across the chunks and are reduced over the 32 pixel lanes ONCE, at the end (DPP butterflies inside the 16-lane rows + one cross-row exchange:
// a fixed order) -- with one reduction per chunk the ds_bpermute traffic of 2 * CPT * 5 shuffles per thread was most of the kernel
template <int CPT>
__global__ __launch_bounds__(256) void k_lc_entry(const LcEntry L) {
    __shared__ float s_a[8][32], s_b[8][32];
    __shared__ int s_ia[8][32], s_ib[8][32];
    const int b = blockIdx.y, px = threadIdx.x & 31, cg = threadIdx.x >> 5;
    const int cpt = (L.C + 7) >> 3, nchunks = (L.hw + 31) >> 5;
    float s1[CPT], s2[CPT];
#pragma unroll
    for (int i = 0; i < CPT; i++) { s1[i] = 0.0f; s2[i] = 0.0f; }
    for (int ck = blockIdx.x; ck < nchunks; ck += gridDim.x) {
        const int p = ck * 32 + px;
        const bool ok = p < L.hw;
        const size_t base = (size_t)b * L.C * L.hw + (ok ? p : 0);
        float x[CPT], g[CPT];
        float mn = __uint_as_float(0x7f800000u), mx = __uint_as_float(0xff800000u);
        int imn = 0x7fffffff, imx = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = cg * cpt + i;
            x[i] = 0.0f; g[i] = 0.0f;
            if (i < cpt && c < L.C) {
                x[i] = L.x[base + (size_t)c * L.hw];
                if (L.gs) g[i] = L.gs[base + (size_t)c * L.hw] * L.scale;
                if (x[i] < mn) { mn = x[i]; imn = c; }   // strict: the first attaining channel of this thread's ascending run
                if (x[i] > mx) { mx = x[i]; imx = c; }
            }
        }
        float dinv = 0.0f, sg = 0.0f, sgs = 0.0f;
        if (L.gs) {  // (workgroup-uniform)
            __syncthreads();  // (the previous chunk's reads of the exchange arrays are done)
            s_a[cg][px] = mn; s_b[cg][px] = mx; s_ia[cg][px] = imn; s_ib[cg][px] = imx;
            __syncthreads();
            mn = s_a[0][px]; imn = s_ia[0][px]; mx = s_b[0][px]; imx = s_ib[0][px];
#pragma unroll
            for (int k = 1; k < 8; k++) {  // channel groups ascend: strict comparison keeps the first attaining channel
                if (s_a[k][px] < mn) { mn = s_a[k][px]; imn = s_ia[k][px]; }
                if (s_b[k][px] > mx) { mx = s_b[k][px]; imx = s_ib[k][px]; }
            }
            const float d = (mx - mn) + 1e-8f;
#pragma unroll
            for (int i = 0; i < CPT; i++) {
                const int c = cg * cpt + i;
                if (i < cpt && c < L.C) {
                    sg += g[i];
                    sgs = fmaf(g[i], (x[i] - mn) / d, sgs);
                }
            }
            __syncthreads();
            s_a[cg][px] = sg; s_b[cg][px] = sgs;
            __syncthreads();
            sg = 0.0f; sgs = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; k++) { sg += s_a[k][px]; sgs += s_b[k][px]; }
            dinv = d;
        }
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = cg * cpt + i;
            if (ok && i < cpt && c < L.C) {
                float t = 0.0f;
                if (L.gs) {
                    t = g[i] / dinv;
                    if (c == imn) t = t - sg / dinv + sgs / dinv;
                    if (c == imx) t = t - sgs / dinv;
                }
                if (L.extra) t += L.extra[base + (size_t)c * L.hw];
                t = x[i] > 0.0f ? t : 0.0f;
                L.dz[base + (size_t)c * L.hw] = t;
                s1[i] += t;
                s2[i] = fmaf(t, L.partner[base + (size_t)c * L.hw], s2[i]);
            }
        }
    }
    // per-channel sums over the 32 pixel lanes of this channel group (one half of a wave): two 16-lane DPP butterflies + the other row
#pragma unroll
    for (int i = 0; i < CPT; i++) {
        float a = mz::butterfly16(s1[i]), bb = mz::butterfly16(s2[i]);
        a += __shfl_xor(a, 16);
        bb += __shfl_xor(bb, 16);
        const int c = cg * cpt + i;
        if (px == 0 && i < cpt && c < L.C) {
            float* d = L.stat_part + (((size_t)b * gridDim.x + blockIdx.x) * L.cpad + c) * 2;
            d[0] = a; d[1] = bb;
        }
    }
}

// k_lc_entry without the normalisation (gs == null) for planes of hw % 4 == 0 positions (the tiled stages of the Atari representation: 2304 / 576 /
// 144): dz = extra [x > 0] + the partial sums, four positions per lane (b128 loads / stores).  grid (nsplit, B) as above, chunks of 128 positions.
template <int CPT>
__global__ __launch_bounds__(256) void k_lc_entry_plain(const LcEntry L) {
    const int b = blockIdx.y, q = threadIdx.x & 31, cg = threadIdx.x >> 5;
    const int cpt = (L.C + 7) >> 3, nq = L.hw >> 2, nchunks = (nq + 31) >> 5;
    float s1[CPT], s2[CPT];
#pragma unroll
    for (int i = 0; i < CPT; i++) { s1[i] = 0.0f; s2[i] = 0.0f; }
    for (int ck = blockIdx.x; ck < nchunks; ck += gridDim.x) {
        const int qi = ck * 32 + q;
        if (qi >= nq) continue;
        const size_t base = (size_t)b * L.C * L.hw + 4 * (size_t)qi;
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = cg * cpt + i;
            if (i < cpt && c < L.C) {
                const size_t o = base + (size_t)c * L.hw;
                const float4 x = *reinterpret_cast<const float4*>(L.x + o), e = *reinterpret_cast<const float4*>(L.extra + o);
                const float4 y = *reinterpret_cast<const float4*>(L.partner + o);
                float4 t;
                t.x = x.x > 0.0f ? e.x : 0.0f; t.y = x.y > 0.0f ? e.y : 0.0f; t.z = x.z > 0.0f ? e.z : 0.0f; t.w = x.w > 0.0f ? e.w : 0.0f;
                *reinterpret_cast<float4*>(L.dz + o) = t;
                s1[i] += (t.x + t.y) + (t.z + t.w);
                s2[i] += fmaf(t.x, y.x, t.y * y.y) + fmaf(t.z, y.z, t.w * y.w);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < CPT; i++) {
        float a = mz::butterfly16(s1[i]), bb = mz::butterfly16(s2[i]);
        a += __shfl_xor(a, 16);
        bb += __shfl_xor(bb, 16);
        const int c = cg * cpt + i;
        if (q == 0 && i < cpt && c < L.C) {
            float* d = L.stat_part + (((size_t)b * gridDim.x + blockIdx.x) * L.cpad + c) * 2;
            d[0] = a; d[1] = bb;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Large images (the Atari representation, network.py:312-353: 96 x 96 -> 48 x 48 -> 24 x 24 before the hidden state's 6 x 6): the convolution
// kernels above hold WHOLE images of at most 240 points.  A large plane is cut into T x T tiles, gathered WITH their one-pixel halo into
// (T + 2) x (T + 2) "images" (zero outside the plane), convolved by the same kernels -- of every tile's output only the inner T x T is
// meaningful -- and scattered back.  The staging transform (BatchNorm + ReLU, BatchNorm backward) is applied by the gather, which is the one
// place that knows which halo positions lie outside the plane (the convolution pads the TRANSFORMED activation with zeros).  A stride-2
// convolution (conv_1, conv_2) is four stride-1 convolutions over the parity planes of its input, x[2 y' + p][2 x' + q]: tap ky of output
// row oy reads input row 2 oy + ky - 1, i.e. the even plane's row oy for ky = 1 and the odd plane's rows oy - 1 / oy for ky = 0 / 2 -- the
// gather takes a stride and an origin, the packed weights of each plane hold its taps and zeros elsewhere.
// ---------------------------------------------------------------------------------------------------------------------------------
struct LcTileGather {
    const float* src0;   // [B][C][srcH][srcW]
    const float* src1;   // IN_BNBWD: y
    const float* coef;   // [3][cpad] or null (IN_IDENT)
    float* dst;          // [B * nty * ntx][C][(Ty + 2) (Tx + 2)]
    int mode;            // IN_IDENT | IN_BNRELU | IN_BNBWD
    int B, C, cpad, H, W;            // the plane that is tiled (a parity plane: H = srcH / 2)
    int srcH, srcW, sy, sx, py, px;  // plane position (y, x) is source element (y * sy + py, x * sx + px)
    int Ty, Tx, nty, ntx;            // tile size (rows x columns of plane positions) and count
    int inner_only;      // 1: the halo ring is written as zeros (the dy operand of the weight gradient: only the tile's own pixels count)
    long long n;         // elements of dst
    int xcd;             // 1: XCD-aware tile order (k_lc_tile_gather)
};
// grid (tiles, chunks of 1024 tile elements): the tile's coordinates are workgroup-uniform (scalar), the element's (c, ly, lx) divide by compile-time
// constants; four elements per thread, their loads issued together (one 4-byte load per thread in flight left the kernel latency-bound at 1.5 TB/s)
template <int TSY, int TSX>
__global__ __launch_bounds__(256) void k_lc_tile_gather(const LcTileGather L) {
    constexpr int ts2 = TSY * TSX;
    // XCD-aware order (round 6): workgroups go to the 8 XCDs round-robin by linear id, and neighbouring tiles share the source plane's cache lines (a
    // 14 x 18 tile's rows are 72 bytes at a 192-byte pitch): in launch order every line was fetched into two or three L2s.  Remapped, XCD k takes
    // the k-th eighth of the tiles -- whole images next to each other.
    int ti = blockIdx.x;
    if (L.xcd && (gridDim.x & 7) == 0) ti = (ti & 7) * (gridDim.x >> 3) + (ti >> 3);
    const int j0 = blockIdx.y * 1024 + threadIdx.x, nel = L.C * ts2;
    const int nt = L.nty * L.ntx, b = ti / nt, t = ti - b * nt, tyi = t / L.ntx, txi = t - tyi * L.ntx;
    const float* s0 = L.src0 + (size_t)b * L.C * L.srcH * L.srcW;
    const float* s1 = L.mode == IN_BNBWD ? L.src1 + (size_t)b * L.C * L.srcH * L.srcW : s0;
    float v[4], v1[4];
    int cc[4];
    bool in[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int j = j0 + u * 256;
        const int c = j / ts2, r = j - c * ts2, ly = r / TSX, lx = r - ly * TSX;
        const int y = tyi * L.Ty - 1 + ly, x = txi * L.Tx - 1 + lx;
        const bool ring = ly == 0 || lx == 0 || ly == TSY - 1 || lx == TSX - 1;
        in[u] = j < nel && y >= 0 && y < L.H && x >= 0 && x < L.W && !(L.inner_only && ring);
        cc[u] = c;
        const size_t si = in[u] ? (size_t)c * L.srcH * L.srcW + (size_t)(y * L.sy + L.py) * L.srcW + (x * L.sx + L.px) : 0;
        v[u] = s0[si];
        v1[u] = L.mode == IN_BNBWD ? s1[si] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int j = j0 + u * 256;
        float o = 0.0f;
        if (in[u]) {
            o = v[u];
            if (L.mode == IN_BNRELU) {
                o = fmaf(L.coef[cc[u]], o, L.coef[L.cpad + cc[u]]);
                o = o > 0.0f ? o : 0.0f;
            } else if (L.mode == IN_BNBWD) {
                o = fmaf(L.coef[cc[u]], o, fmaf(L.coef[L.cpad + cc[u]], v1[u], L.coef[2 * L.cpad + cc[u]]));
            }
        }
        if (j < nel) L.dst[(size_t)ti * nel + j] = o;
    }
}

// inner T x T of every tile -> the plane (a parity plane of dst when sy = 2), + skip; optional forward statistics (sum v, sum v^2) per
// (workgroup, channel).  workgroup = one image x 32 plane positions; thread (position, channel group of 8)
struct LcTileScatter {
    const float* src;    // tiles [B * nty * ntx][C][(T + 2)^2]
    float* dst;          // [B][C][dstH][dstW]
    const float* skip;   // like dst, or null
    float* stat_part;    // [B * chunks][cpad][2] or null
    int B, C, cpad, H, W, dstH, dstW, sy, sx, py, px, Ty, Tx, nty, ntx;
    int src_inner;       // 1: src holds the INNER Ty x Tx positions of every tile only (the output of a halo_in convolution): [tiles][C][Ty Tx]
};
template <int CPT>
__global__ __launch_bounds__(256) void k_lc_tile_scatter(const LcTileScatter L) {
    const int b = blockIdx.y, lp = threadIdx.x & 31, cg = threadIdx.x >> 5, p = blockIdx.x * 32 + lp;
    const int cpt = (L.C + 7) >> 3, TSX = L.Tx + 2, ts2h = (L.Ty + 2) * TSX;
    const bool ok = p < L.H * L.W;
    const int y = ok ? p / L.W : 0, x = ok ? p - (p / L.W) * L.W : 0;
    const int tyi = y / L.Ty, txi = x / L.Tx, ly = y - tyi * L.Ty + 1, lx = x - txi * L.Tx + 1;
    const int ts2 = L.src_inner ? L.Ty * L.Tx : ts2h;
    const size_t sbase = ((size_t)(b * L.nty + tyi) * L.ntx + txi) * L.C * ts2 + (L.src_inner ? (size_t)(ly - 1) * L.Tx + (lx - 1) : (size_t)ly * TSX + lx);
    const size_t dbase = (size_t)b * L.C * L.dstH * L.dstW + (size_t)(y * L.sy + L.py) * L.dstW + (x * L.sx + L.px);
#pragma unroll
    for (int i = 0; i < CPT; i++) {
        const int c = cg * cpt + i;
        float v = 0.0f;
        if (ok && i < cpt && c < L.C) {
            v = L.src[sbase + (size_t)c * ts2];
            const size_t di = dbase + (size_t)c * L.dstH * L.dstW;
            if (L.skip) v += L.skip[di];
            L.dst[di] = v;
        }
        if (L.stat_part) {  // pivoted (see k_lc_conv's ST_FWD): the chunk's first position is the pivot (always a valid one)
            const float pvt = __shfl(v, threadIdx.x & 32);
            const float u = ok ? v - pvt : 0.0f;
            float a = mz::butterfly16(u), bb = mz::butterfly16(u * u);
            a += __shfl_xor(a, 16);
            bb += __shfl_xor(bb, 16);
            if (lp == 0 && i < cpt && c < L.C) {
                const int left = L.H * L.W - (int)blockIdx.x * 32;
                *reinterpret_cast<float4*>(L.stat_part + (((size_t)b * gridDim.x + blockIdx.x) * L.cpad + c) * 4) =
                    make_float4(pvt, a, bb, (float)(left < 32 ? left : 32));
            }
        }
    }
}

// nn.AvgPool2d(3, 2, 1), count_include_pad (network.py:337,342), and its backward (every input pixel collects 1 / 9 of the outputs whose window covers it)
__global__ __launch_bounds__(256) void k_lc_pool_fwd(const float* in, float* out, long long n, int ih, int iw, int oh, int ow) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
    const float* p = in + (i / ((long long)ow * oh)) * ih * iw;
    float acc = 0.0f;
    for (int ky = 0; ky < 3; ky++)
        for (int kx = 0; kx < 3; kx++) {
            const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
            if (iy >= 0 && iy < ih && ix >= 0 && ix < iw) acc = acc + p[iy * iw + ix];
        }
    out[i] = acc / 9.0f;
}
__global__ __launch_bounds__(256) void k_lc_pool_bwd(const float* gout, float* gin, long long n, int ih, int iw, int oh, int ow) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // over the INPUT pixels
    if (i >= n) return;
    const int ix = (int)(i % iw), iy = (int)((i / iw) % ih);
    const float* g = gout + (i / ((long long)iw * ih)) * oh * ow;
    float acc = 0.0f;
    for (int oy = (iy + 0) / 2; oy <= (iy + 1) / 2; oy++)      // 2 oy - 1 <= iy <= 2 oy + 1
        for (int ox = (ix + 0) / 2; ox <= (ix + 1) / 2; ox++)
            if (oy < oh && ox < ow) acc = acc + g[oy * ow + ox];
    gin[i] = acc / 9.0f;
}
// dz = g [x > 0] for a ReLU WITHOUT BatchNorm (conv_1 / conv_2 of the Atari representation, network.py:345,347); x = relu(y) materialised
__global__ __launch_bounds__(256) void k_lc_relu_bwd(const float* g, const float* x, float* dz, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dz[i] = x[i] > 0.0f ? g[i] : 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Batch plumbing: replay rows -> dense observations and actions (replay.py:27-32 `Transition` storages addressed by row index)
// ---------------------------------------------------------------------------------------------------------------------------------
struct LcBatch {
    const void* state;     // [cap][C0 * hw] float32 or int8
    const void* action;    // [cap][K] int8 or int16
    const float* pi;       // [cap][K][A]
    const float* value;    // [cap][K]
    const float* reward;   // [cap][K]
    const int64_t* idx;    // [B]
    const float* w;        // [B]
    float* prio;           // [B]
    int B, state_i8, action_bytes, K, A, in_dim;
};
__global__ __launch_bounds__(256) void k_lc_gather(const LcBatch bt, float* obs, int* act) {
    const int b = blockIdx.y;
    const int64_t row = bt.idx[b];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < bt.in_dim; i += gridDim.x * 256)
        obs[(size_t)b * bt.in_dim + i] = bt.state_i8 ? (float)reinterpret_cast<const int8_t*>(bt.state)[row * bt.in_dim + i]
                                                    : reinterpret_cast<const float*>(bt.state)[row * bt.in_dim + i];
    if (blockIdx.x == 0 && threadIdx.x < bt.K) {
        const int k = threadIdx.x;
        const int a = bt.action_bytes == 2 ? (int)reinterpret_cast<const int16_t*>(bt.action)[row * bt.K + k]
                                           : (int)reinterpret_cast<const int8_t*>(bt.action)[row * bt.K + k];
        act[(size_t)k * bt.B + b] = a;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Heads (network.py:424-430 reward, :472-486 policy / value): 1x1 conv (P -> oc) + BatchNorm2d(oc) + ReLU + Flatten + Linear(oc * hw -> n_out).
// A "group" is one application: (unroll step t, head).  All groups run side by side (grid.y): the towers' outputs are saved, and a head's
// loss gradient is known as soon as its logits are.
//   k_lch_conv   u[o][p] = sum_c w1[o][c] F[b][c][p]; partial sums (sum u, sum u^2) per (group, image)
//   k_lch_bn     batch statistics per head (the K steps of a head update its running statistics in step order)
//   k_lch_loss   feat = relu(a u + b); logits; loss; dlogits; dfeat; dz = dfeat [feat > 0]; partial sums (sum dz, sum dz u); priorities
//   k_lch_bnb    BatchNorm backward coefficients, dgamma / dbeta (summed over the steps), the reported loss
//   k_lch_dx     dF[b][c][p] = sum over the heads on F, o: w1[o][c] du[o][p],  du = c1 dz + c2 u + c3
//   k_lch_dw1 / k_lch_dlin   weight gradients (sums over steps and batch in fixed order)
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int LCH_MAXOC = 2;
struct LchHead {           // one of reward / policy / value
    int oc, n_out, kind;   // kind 0: squared error (n_out == 1), 1: soft-target cross entropy (policy)
    int w1_off, gamma_off, beta_off, lw_off, lb_off;  // offsets into params / grads
    int rm_off, rv_off, nbt_idx;                      // running statistics
};
struct LchGroup {
    const float* F;        // [B][P][hw] tower output this application reads
    int head, t;
};
struct LchArgs {
    LchHead head[3];
    const LchGroup* groups;  // [ngroups] device
    int ngroups, K, B, P, hw, A;
    const float* params;
    float* grads;
    float* running;        // flat running statistics
    int64_t* nbt;
    const float* lwT;      // per head: transposed Linear weights [oc * hw][n_out] at lwT_off[head]
    int lwT_off[3];
    float* u;              // [ngroups][B][2][hw]
    float* dzb;            // [ngroups][B][2][hw]
    float* feat;           // [ngroups][B][2 * hw]
    float* dlogit;         // [ngroups][B][n_max]
    float* spart;          // [ngroups][B][2 oc][2] forward partials (pivoted: sum (u - p), sum (u - p)^2), then backward partials
    float* spiv;           // [ngroups][B][2 oc] the forward partials' pivots
    float* coef;           // [ngroups][2 oc][5]: a, b (forward); c1, c2, c3 (backward)
    float* save;           // [ngroups][2 oc][2]: mean, invstd
    float* lpart;          // [ngroups][B] weighted loss terms
    int n_max;
    LcBatch bt;
    float* loss;           // [1]
    float* wpart;          // [K][hp_total]: per-step partial weight gradients of the heads (k_lch_dw1 / k_lch_dlin), summed in step order by k_lch_wsum
    int hp_off[3], hp_total;  // per head: [w1 (oc * P)] [lw (n_out * oc * hw)] [lb (n_out)]
};

__global__ __launch_bounds__(256) void k_lch_conv(const LchArgs A) {
    __shared__ float s_w[LCH_MAXOC][1024];
    __shared__ float s_red[2 * LCH_MAXOC][4];
    const int g = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
    const LchGroup G = A.groups[g];
    const LchHead H = A.head[G.head];
    for (int i = tid; i < H.oc * A.P; i += 256) s_w[i / A.P][i % A.P] = A.params[H.w1_off + i];
    __syncthreads();
    const float* F = G.F + (size_t)b * A.P * A.hw;
    // one position per thread (hw <= 240 < 256 threads: mzlc_create's limit); the image's statistics are summed around a pivot -- u at position 0,
    // published by thread 0 -- see k_lc_conv's ST_FWD
    __shared__ float s_piv[LCH_MAXOC];
    const int p = tid;
    float acc[LCH_MAXOC] = {0.f, 0.f};
    if (p < A.hw) {
        for (int c = 0; c < A.P; c++) {
            const float f = F[(size_t)c * A.hw + p];
#pragma unroll
            for (int o = 0; o < LCH_MAXOC; o++)
                if (o < H.oc) acc[o] = fmaf(s_w[o][c], f, acc[o]);
        }
#pragma unroll
        for (int o = 0; o < LCH_MAXOC; o++)
            if (o < H.oc) A.u[(((size_t)g * A.B + b) * LCH_MAXOC + o) * A.hw + p] = acc[o];
    }
    if (tid == 0) {
#pragma unroll
        for (int o = 0; o < LCH_MAXOC; o++) s_piv[o] = acc[o];
    }
    __syncthreads();
    float pvt[LCH_MAXOC], s1[LCH_MAXOC], s2[LCH_MAXOC];
#pragma unroll
    for (int o = 0; o < LCH_MAXOC; o++) {
        pvt[o] = s_piv[o];
        const float d = (p < A.hw && o < H.oc) ? acc[o] - pvt[o] : 0.0f;
        s1[o] = d;
        s2[o] = d * d;
    }
    // block sums in a fixed order: wave butterflies, then the four waves in order
#pragma unroll
    for (int o = 0; o < LCH_MAXOC; o++) {
        for (int m = 32; m >= 1; m >>= 1) { s1[o] += __shfl_xor(s1[o], m); s2[o] += __shfl_xor(s2[o], m); }
        if ((tid & 63) == 0) { s_red[2 * o][tid >> 6] = s1[o]; s_red[2 * o + 1][tid >> 6] = s2[o]; }
    }
    __syncthreads();
    if (tid < 2 * H.oc) {
        const float s = ((s_red[tid][0] + s_red[tid][1]) + s_red[tid][2]) + s_red[tid][3];
        A.spart[(((size_t)g * A.B + b) * LCH_MAXOC + (tid >> 1)) * 2 + (tid & 1)] = s;
        if ((tid & 1) == 0) A.spiv[((size_t)g * A.B + b) * LCH_MAXOC + (tid >> 1)] = pvt[tid >> 1];
    }
}

// forward statistics of a head from its pivoted per-image partials (n = hw positions each): see lc_pivot_sums
__device__ __forceinline__ void lch_wave_pivot_sums(const float* spart, const float* spiv, int g, int B, int o, double n, double& nmean, double& nvar) {
    const int lane = threadIdx.x & 63;
    double T1 = 0.0, T2 = 0.0, U1 = 0.0, U2 = 0.0, V = 0.0;
    for (int i = lane; i < B; i += 64) {
        const size_t e = ((size_t)g * B + i) * LCH_MAXOC + o;
        const double a = spart[e * 2], b = spart[e * 2 + 1], p = spiv[e];
        T1 += a; T2 += b; U1 += n * p; U2 += n * p * p; V += p * a;
    }
    for (int m = 32; m >= 1; m >>= 1) { T1 += __shfl_xor(T1, m); T2 += __shfl_xor(T2, m); U1 += __shfl_xor(U1, m); U2 += __shfl_xor(U2, m); V += __shfl_xor(V, m); }
    const double N = n * (double)B;
    nmean = U1 + T1;
    const double mean = nmean / N;
    nvar = T2 + 2.0 * V - 2.0 * mean * T1 + U2 - 2.0 * mean * U1 + mean * mean * N;
}

// one WAVE per (head, plane): the K applications of a head in step order (their running-statistics updates are sequential); the batch sum is
// 64 strided float64 partials + a fixed-order butterfly
__device__ __forceinline__ void lch_wave_sums(const float* spart, int g, int B, int o, double& s1, double& s2) {
    const int lane = threadIdx.x & 63;
    double a = 0.0, b = 0.0;
    for (int i = lane; i < B; i += 64) {
        const float* p = spart + (((size_t)g * B + i) * LCH_MAXOC + o) * 2;
        a += (double)p[0]; b += (double)p[1];
    }
    for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
    s1 = a; s2 = b;
}
__device__ __forceinline__ int lch_group_of(const LchArgs& A, int hd, int t) {
    int g = -1;
    for (int k = 0; k < A.ngroups; k++)
        if (A.groups[k].head == hd && A.groups[k].t == t) g = k;
    return g;
}
__global__ __launch_bounds__(64 * 3 * LCH_MAXOC) void k_lch_bn(const LchArgs A) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hd = wave / LCH_MAXOC, o = wave % LCH_MAXOC;
    if (hd >= 3 || o >= A.head[hd].oc) return;
    const LchHead H = A.head[hd];
    const double M = (double)A.B * A.hw;
    for (int t = 0; t < A.K; t++) {
        const int g = lch_group_of(A, hd, t);
        if (g < 0) continue;
        double nmean, nvar;
        lch_wave_pivot_sums(A.spart, A.spiv, g, A.B, o, (double)A.hw, nmean, nvar);
        if (lane != 0) continue;
        const double mean = nmean / M;
        double var = nvar / M;
        var = var < 0.0 ? 0.0 : var;
        const float invstd = (float)(1.0 / sqrt(var + 1e-5));
        const float a = A.params[H.gamma_off + o] * invstd;
        float* cf = A.coef + ((size_t)g * LCH_MAXOC + o) * 5;
        cf[0] = a;
        cf[1] = A.params[H.beta_off + o] - (float)mean * a;
        A.save[((size_t)g * LCH_MAXOC + o) * 2] = (float)mean;
        A.save[((size_t)g * LCH_MAXOC + o) * 2 + 1] = invstd;
        const double unb = M > 1.0 ? var * M / (M - 1.0) : var;
        A.running[H.rm_off + o] = (float)(0.9 * (double)A.running[H.rm_off + o] + 0.1 * mean);
        A.running[H.rv_off + o] = (float)(0.9 * (double)A.running[H.rv_off + o] + 0.1 * unb);
        if (o == 0) A.nbt[H.nbt_idx] += 1;
    }
}

__global__ __launch_bounds__(256) void k_lch_loss(const LchArgs A) {
    extern __shared__ float lsm[];
    const int g = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
    const LchGroup G = A.groups[g];
    const LchHead H = A.head[G.head];
    const int nf = H.oc * A.hw;
    float* s_feat = lsm;               // [nf]
    float* s_lg = s_feat + nf;         // [n_out]
    float* s_dl = s_lg + H.n_out;      // [n_out]
    float* s_red = s_dl + H.n_out;     // [16]
    const float* u = A.u + ((size_t)g * A.B + b) * LCH_MAXOC * A.hw;
    const float* cf = A.coef + (size_t)g * LCH_MAXOC * 5;
    for (int k = tid; k < nf; k += 256) {
        const int o = k / A.hw, p = k - o * A.hw;
        const float z = fmaf(cf[o * 5], u[o * A.hw + p], cf[o * 5 + 1]);
        const float f = z > 0.0f ? z : 0.0f;
        s_feat[k] = f;
        A.feat[((size_t)g * A.B + b) * LCH_MAXOC * A.hw + k] = f;
    }
    __syncthreads();
    const float* lwT = A.lwT + A.lwT_off[G.head];
    for (int n = tid; n < H.n_out; n += 256) {
        float a = A.params[H.lb_off + n];
        for (int k = 0; k < nf; k++) a = fmaf(s_feat[k], lwT[(size_t)k * H.n_out + n], a);
        s_lg[n] = a;
    }
    __syncthreads();
    const int64_t row = A.bt.idx[b];
    const float wgt = A.bt.w[b];
    const float gscale = wgt / ((float)A.B * (float)A.K);  // mean over the batch, 1 / K on the gradient (pipeline.py:597-600)
    if (H.kind == 0) {  // squared error (pipeline.py:625: F.mse_loss, reduction none)
        if (tid == 0) {
            const float target = (G.head == 0 ? A.bt.reward : A.bt.value)[row * A.K + G.t];
            const float d = s_lg[0] - target;
            A.lpart[(size_t)g * A.B + b] = d * d * wgt;
            s_dl[0] = 2.0f * d * gscale;
            if (G.head == 2 && G.t == 0) A.bt.prio[b] = fabsf(d);  // pipeline.py:603-609
        }
    } else {            // cross entropy against soft targets (pipeline.py:629): the policy's visit distribution, or (kind 2) the 2-hot projection
                        // of the transformed scalar target onto the support (util.py:48-59,96-116: signed_hyperbolic, clamp, two neighbouring bins)
        float* s_tg = s_red + 16;  // [n_out] (kind 2)
        const float* pi = A.bt.pi + ((size_t)row * A.K + G.t) * A.A;
        float tscalar = 0.0f;
        if (H.kind == 2) {
            tscalar = (G.head == 0 ? A.bt.reward : A.bt.value)[row * A.K + G.t];
            for (int n = tid; n < H.n_out; n += 256) s_tg[n] = 0.0f;
            __syncthreads();
            if (tid == 0) {
                const float half = (float)((H.n_out - 1) / 2);
                const float ax = fabsf(tscalar), sg = tscalar > 0.0f ? 1.0f : (tscalar < 0.0f ? -1.0f : 0.0f);
                float z = sg * (sqrtf(ax + 1.0f) - 1.0f) + 0.001f * tscalar;          // signed_hyperbolic, util.py:20-22
                z = z < -half ? -half : (z > half ? half : z);
                const float span = 2.0f * half, pos = (z + half) / span * (float)(H.n_out - 1);
                const float lo = floorf(pos), hi = ceilf(pos);
                const float slo = lo / ((float)H.n_out - 1.0f) * span - half, shi = hi / ((float)H.n_out - 1.0f) * span - half;
                const float wlo = (shi - z) / (shi - slo + 1e-5f);
                s_tg[(int)lo] += wlo;
                s_tg[(int)hi] += 1.0f - wlo;
            }
            __syncthreads();
            pi = s_tg;
        }
        float mx = __uint_as_float(0xff800000u);
        for (int n = tid; n < H.n_out; n += 256) mx = s_lg[n] > mx ? s_lg[n] : mx;
        for (int m = 32; m >= 1; m >>= 1) { const float o = __shfl_xor(mx, m); mx = o > mx ? o : mx; }
        if ((tid & 63) == 0) s_red[tid >> 6] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        float se = 0.0f, sp = 0.0f, spl = 0.0f, sev = 0.0f;
        const float half_s = (float)((H.n_out - 1) / 2);
        for (int n = tid; n < H.n_out; n += 256) {
            const float e = expf(s_lg[n] - mx);
            se += e;
            sp += pi[n];
            spl = fmaf(pi[n], s_lg[n] - mx, spl);
            sev = fmaf(e, (float)n - half_s, sev);  // (kind 2: expectation over the support, for the priority)
        }
        for (int m = 32; m >= 1; m >>= 1) { se += __shfl_xor(se, m); sp += __shfl_xor(sp, m); spl += __shfl_xor(spl, m); sev += __shfl_xor(sev, m); }
        __syncthreads();
        if ((tid & 63) == 0) { s_red[tid >> 6] = se; s_red[4 + (tid >> 6)] = sp; s_red[8 + (tid >> 6)] = spl; s_red[12 + (tid >> 6)] = sev; }
        __syncthreads();
        se = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
        sp = ((s_red[4] + s_red[5]) + s_red[6]) + s_red[7];
        spl = ((s_red[8] + s_red[9]) + s_red[10]) + s_red[11];
        sev = ((s_red[12] + s_red[13]) + s_red[14]) + s_red[15];
        const float lse = logf(se);
        if (tid == 0) {
            A.lpart[(size_t)g * A.B + b] = (sp * lse - spl) * wgt;   // -sum pi (l - mx - lse)
            // priority |v_0 - z_0| in scalar space (pipeline.py:603-609): logits_to_transformed_expected_value (util.py:70-93)
            if (H.kind == 2 && G.head == 2 && G.t == 0) A.bt.prio[b] = fabsf(mz::signed_parabolic(sev / se) - tscalar);
        }
        for (int n = tid; n < H.n_out; n += 256) s_dl[n] = (expf(s_lg[n] - mx) / se * sp - pi[n]) * gscale;
    }
    __syncthreads();
    for (int n = tid; n < H.n_out; n += 256) A.dlogit[((size_t)g * A.B + b) * A.n_max + n] = s_dl[n];
    // dfeat[k] = sum_n lw[n][k] dlogit[n]; dz = dfeat [feat > 0]; partial sums of the BatchNorm backward
    const float* lw = A.params + H.lw_off;
    float s1[LCH_MAXOC] = {0.f, 0.f}, s2[LCH_MAXOC] = {0.f, 0.f};
    for (int k = tid; k < nf; k += 256) {
        float a = 0.0f;
        for (int n = 0; n < H.n_out; n++) a = fmaf(lw[(size_t)n * nf + k], s_dl[n], a);
        const float dz = s_feat[k] > 0.0f ? a : 0.0f;
        const int o = k / A.hw, p = k - o * A.hw;
        A.dzb[(((size_t)g * A.B + b) * LCH_MAXOC + o) * A.hw + p] = dz;
#pragma unroll
        for (int oo = 0; oo < LCH_MAXOC; oo++)
            if (oo == o) { s1[oo] += dz; s2[oo] = fmaf(dz, u[o * A.hw + p], s2[oo]); }
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < LCH_MAXOC; o++) {
        for (int m = 32; m >= 1; m >>= 1) { s1[o] += __shfl_xor(s1[o], m); s2[o] += __shfl_xor(s2[o], m); }
        if ((tid & 63) == 0) { s_red[(2 * o) * 4 + (tid >> 6)] = s1[o]; s_red[(2 * o + 1) * 4 + (tid >> 6)] = s2[o]; }
    }
    __syncthreads();
    if (tid < 2 * H.oc) {
        const float s = ((s_red[tid * 4] + s_red[tid * 4 + 1]) + s_red[tid * 4 + 2]) + s_red[tid * 4 + 3];
        A.spart[(((size_t)g * A.B + b) * LCH_MAXOC + (tid >> 1)) * 2 + (tid & 1)] = s;
    }
}

// BatchNorm backward of the heads + the reported loss (one workgroup: waves 0 .. 3 * LCH_MAXOC - 1 own one (head, plane) each, the last
// wave adds up the loss terms)
__global__ __launch_bounds__(64 * (3 * LCH_MAXOC + 1)) void k_lch_bnb(const LchArgs A) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hd = wave / LCH_MAXOC, o = wave % LCH_MAXOC;
    if (wave == 3 * LCH_MAXOC) {  // loss = mean_b w_b sum_t (reward + value + policy) (pipeline.py:594-597)
        double s = 0.0;
        for (int i = lane; i < A.ngroups * A.B; i += 64) s += (double)A.lpart[i];
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        if (lane == 0) *A.loss = (float)(s / (double)A.B);
        return;
    }
    if (hd >= 3 || o >= A.head[hd].oc) return;
    const LchHead H = A.head[hd];
    const double M = (double)A.B * A.hw;
    double dg_sum = 0.0, db_sum = 0.0;
    for (int t = 0; t < A.K; t++) {
        const int g = lch_group_of(A, hd, t);
        if (g < 0) continue;
        double s1, s2;
        lch_wave_sums(A.spart, g, A.B, o, s1, s2);
        const double mean = A.save[((size_t)g * LCH_MAXOC + o) * 2], invstd = A.save[((size_t)g * LCH_MAXOC + o) * 2 + 1];
        const double gam = A.params[H.gamma_off + o];
        const double dgam = (s2 - mean * s1) * invstd, c1 = gam * invstd;
        if (lane == 0) {
            float* cf = A.coef + ((size_t)g * LCH_MAXOC + o) * 5;
            cf[2] = (float)c1;
            cf[3] = (float)(-c1 * invstd * dgam / M);
            cf[4] = (float)(-c1 * s1 / M + c1 * mean * invstd * dgam / M);
        }
        dg_sum += dgam; db_sum += s1;
    }
    if (lane == 0) {
        A.grads[H.gamma_off + o] = (float)dg_sum;
        A.grads[H.beta_off + o] = (float)db_sum;
    }
}

// gradient wrt a tower output from the heads reading it: grid (B, K, 2): z = 0 the prediction tower's f_t (policy + value), z = 1 the dynamics
// tower's raw g_t (reward)
struct LchDx {
    float* out[2];  // [K][B][P][hw] each
};
__global__ __launch_bounds__(256) void k_lch_dx(const LchArgs A, const LchDx D) {
    __shared__ float s_w[3][LCH_MAXOC][1024];
    const int b = blockIdx.x, t = blockIdx.y, which = blockIdx.z, tid = threadIdx.x;
    int gs[3], ng = 0;
    for (int k = 0; k < A.ngroups; k++)
        if (A.groups[k].t == t && ((A.groups[k].head == 0) == (which == 1))) gs[ng++] = k;
    int noc[3] = {0, 0, 0};
    for (int e = 0; e < ng; e++) {
        const LchHead H = A.head[A.groups[gs[e]].head];
        noc[e] = H.oc;
        for (int i = tid; i < H.oc * A.P; i += 256) s_w[e][i / A.P][i % A.P] = A.params[H.w1_off + i];
    }
    __syncthreads();
    float* out = D.out[which] + ((size_t)t * A.B + b) * A.P * A.hw;
    for (int p = tid; p < A.hw; p += 256) {
        float du[3][LCH_MAXOC];
        for (int e = 0; e < ng; e++) {
            const int g = gs[e];
            const LchHead H = A.head[A.groups[g].head];
            for (int o = 0; o < LCH_MAXOC; o++) {
                du[e][o] = 0.0f;
                if (o < H.oc) {
                    const float* cf = A.coef + ((size_t)g * LCH_MAXOC + o) * 5;
                    const size_t ix = (((size_t)g * A.B + b) * LCH_MAXOC + o) * A.hw + p;
                    du[e][o] = fmaf(cf[2], A.dzb[ix], fmaf(cf[3], A.u[ix], cf[4]));
                }
            }
        }
        for (int c = 0; c < A.P; c++) {
            float a = 0.0f;
            for (int e = 0; e < ng; e++)
                for (int o = 0; o < LCH_MAXOC; o++)
                    if (o < noc[e]) a = fmaf(s_w[e][o][c], du[e][o], a);  // (planes a head does not have were never staged: LDS garbage, possibly NaN)
            out[(size_t)c * A.hw + p] = a;
        }
    }
}

// dw1_t[o][c] = sum_b sum_p du[t][b][o][p] F_t[b][c][p]: grid (P, 2, K): y = 0 the heads on the prediction tower's output (policy's two planes +
// value's one: F is read ONCE for the three), y = 1 the reward head on the dynamics tower's raw output; z = unroll step (partials, k_lch_wsum)
__global__ __launch_bounds__(256) void k_lch_dw1(const LchArgs A) {
    __shared__ float s_red[3][4];
    const int c = blockIdx.x, which = blockIdx.y, t = blockIdx.z, tid = threadIdx.x;
    int hs[3], os_[3], gs[3], ns = 0;
    float c1[3], c2[3], c3[3];
    for (int hd = 0; hd < 3; hd++)
        if ((hd == 0) == (which == 1))
            for (int o = 0; o < A.head[hd].oc; o++) { hs[ns] = hd; os_[ns] = o; ns++; }
    for (int e = 0; e < ns; e++) {
        gs[e] = lch_group_of(A, hs[e], t);
        const float* cf = A.coef + ((size_t)gs[e] * LCH_MAXOC + os_[e]) * 5;
        c1[e] = cf[2]; c2[e] = cf[3]; c3[e] = cf[4];
    }
    const float* F = A.groups[gs[0]].F;
    float s[3] = {0.f, 0.f, 0.f};
    const int n = A.B * A.hw;
    for (int i = tid; i < n; i += 256) {
        const int b = i / A.hw, p = i - b * A.hw;
        const float f = F[((size_t)b * A.P + c) * A.hw + p];
        for (int e = 0; e < ns; e++) {
            const size_t ix = (((size_t)gs[e] * A.B + b) * LCH_MAXOC + os_[e]) * A.hw + p;
            s[e] = fmaf(fmaf(c1[e], A.dzb[ix], fmaf(c2[e], A.u[ix], c3[e])), f, s[e]);
        }
    }
    for (int e = 0; e < ns; e++) {
        float v = s[e];
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if ((tid & 63) == 0) s_red[e][tid >> 6] = v;
    }
    __syncthreads();
    if (tid == 0)
        for (int e = 0; e < ns; e++)
            A.wpart[(size_t)t * A.hp_total + A.hp_off[hs[e]] + os_[e] * A.P + c] = ((s_red[e][0] + s_red[e][1]) + s_red[e][2]) + s_red[e][3];
}

// dlw_t[n][k] = sum_b dlogit[n] feat[k]; dlb_t[n] = sum_b dlogit[n].  grid (ceil(nf / 256), ceil(n_max / LCH_DLN), 3 K): thread = feature k x
// LCH_DLN outputs n of one (head, step): a feature value is loaded once for LCH_DLN products, the dlogit values are wave-uniform loads
constexpr int LCH_DLN = 4;
__global__ __launch_bounds__(256) void k_lch_dlin(const LchArgs A) {
    const int hd = blockIdx.z % 3, t = blockIdx.z / 3, n0 = blockIdx.y * LCH_DLN;
    const LchHead H = A.head[hd];
    if (n0 >= H.n_out) return;
    const int nf = H.oc * A.hw, k = blockIdx.x * 256 + threadIdx.x;
    const int kc = k < nf ? k : nf - 1;
    const int g = lch_group_of(A, hd, t);
    float s[LCH_DLN], sb[LCH_DLN];
#pragma unroll
    for (int u = 0; u < LCH_DLN; u++) { s[u] = 0.0f; sb[u] = 0.0f; }
    for (int b = 0; b < A.B; b++) {
        const float f = A.feat[((size_t)g * A.B + b) * LCH_MAXOC * A.hw + kc];
        const float* dl = A.dlogit + ((size_t)g * A.B + b) * A.n_max + n0;
#pragma unroll
        for (int u = 0; u < LCH_DLN; u++) {
            const float d = n0 + u < H.n_out ? dl[u] : 0.0f;
            s[u] = fmaf(d, f, s[u]);
            sb[u] += d;
        }
    }
    float* part = A.wpart + (size_t)t * A.hp_total + A.hp_off[hd] + H.oc * A.P;
#pragma unroll
    for (int u = 0; u < LCH_DLN; u++) {
        if (n0 + u >= H.n_out) continue;
        if (k < nf) part[(size_t)(n0 + u) * nf + k] = s[u];
        if (k == 0) part[(size_t)H.n_out * nf + n0 + u] = sb[u];
    }
}

// the heads' weight gradients: per-step partials summed in step order
__global__ __launch_bounds__(256) void k_lch_wsum(const LchArgs A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.hp_total) return;
    int hd = 2;
    if (i < A.hp_off[1]) hd = 0;
    else if (i < A.hp_off[2]) hd = 1;
    const LchHead H = A.head[hd];
    const int l = i - A.hp_off[hd], nf = H.oc * A.hw, nw1 = H.oc * A.P, nlw = H.n_out * nf;
    float s = 0.0f;
    for (int t = 0; t < A.K; t++) s += A.wpart[(size_t)t * A.hp_total + i];
    if (l < nw1) A.grads[H.w1_off + l] = s;
    else if (l < nw1 + nlw) A.grads[H.lw_off + (l - nw1)] = s;
    else A.grads[H.lb_off + (l - nw1 - nlw)] = s;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Operand copies and the optimizer
// ---------------------------------------------------------------------------------------------------------------------------------
struct LcPackJob {
    int w_off;             // master weight [cout][cin][3][3] in params
    int cout, cin, cin_d;  // cin_d: input channels that receive a gradient (the hidden part), 0: no dgrad copy
    int f_off, d_off;      // float offsets of the forward / dgrad copies in the packed buffer
    int n_cb, co_tiles;    // forward: 16-blocks of cin, tiles of cout
    int n_cb_d, co_tiles_d;  // dgrad: 16-blocks of cout, tiles of cin_d
};
__global__ __launch_bounds__(256) void k_lc_pack(const LcPackJob* jobs, const float* params, float* packed) {
    const LcPackJob J = jobs[blockIdx.y];
    const int nf = J.co_tiles * J.n_cb * 9 * 256, nd = J.co_tiles_d * J.n_cb_d * 9 * 256;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nf + nd; e += gridDim.x * 256) {
        const bool fwd = e < nf;
        const int x = fwd ? e : e - nf;
        const int i = x & 3, lane = (x >> 2) & 63, rest = x >> 8;
        const int tap = rest % 9, cb = (rest / 9) % (fwd ? J.n_cb : J.n_cb_d), ct = rest / (9 * (fwd ? J.n_cb : J.n_cb_d));
        const int q = lane >> 4, jj = lane & 15;
        float v = 0.0f;
        if (fwd) {
            const int co = 16 * ct + jj, ci = 16 * cb + 4 * i + q;
            if (co < J.cout && ci < J.cin) v = params[J.w_off + ((size_t)co * J.cin + ci) * 9 + tap];
            packed[J.f_off + x] = v;
        } else {  // "output" channel = ci of the forward layer, "input" = co, taps flipped
            const int ci = 16 * ct + jj, co = 16 * cb + 4 * i + q;
            if (co < J.cout && ci < J.cin_d) v = params[J.w_off + ((size_t)co * J.cin + ci) * 9 + (8 - tap)];
            packed[J.d_off + x] = v;
        }
    }
}
// operand copy of one parity plane of a stride-2 conv (see LcTileGather): tap t of the stride-1 kernel the tiles are convolved with is weight tap
// tapmap[t] (or 0 where the plane has no such tap).  transpose: the data gradient's copy ("output" channel = the layer's input channel)
struct LcPackPar {
    int w_off, cout, cin, dst_off, n_cb, co_tiles, transpose;
    int compact;  // 1: only the plane's own taps are stored, [co tile][block][k-th tap of the plane] (k_lc_conv's TAPMASK builds); 0: nine taps, zeros elsewhere
    signed char tapmap[9];
};
__global__ __launch_bounds__(256) void k_lc_pack_par(const LcPackPar* jobs, const float* params, float* packed) {
    const LcPackPar J = jobs[blockIdx.y];
    int nt = 9, taps[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
    if (J.compact) {
        nt = 0;
        for (int t = 0; t < 9; t++)
            if (J.tapmap[t] >= 0) taps[nt++] = t;
    }
    const int n = J.co_tiles * J.n_cb * nt * 256;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < n; x += gridDim.x * 256) {
        const int i = x & 3, lane = (x >> 2) & 63, rest = x >> 8;
        const int tap = taps[rest % nt], cb = (rest / nt) % J.n_cb, ct = rest / (nt * J.n_cb);
        const int q = lane >> 4, jj = lane & 15, src = J.tapmap[tap];
        const int o = 16 * ct + jj, k = 16 * cb + 4 * i + q;   // output channel of this conv, reduction channel
        const int co = J.transpose ? k : o, ci = J.transpose ? o : k;
        float v = 0.0f;
        if (src >= 0 && co < J.cout && ci < J.cin) v = params[J.w_off + ((size_t)co * J.cin + ci) * 9 + src];
        packed[J.dst_off + x] = v;
    }
}
// transposed Linear weights of the heads: lwT[k][n] = lw[n][k]
__global__ __launch_bounds__(256) void k_lc_pack_lin(const float* params, float* lwT, int lw_off, int n_out, int nf) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out * nf) return;
    const int n = i / nf, k = i - n * nf;
    lwT[(size_t)k * n_out + n] = params[lw_off + i];
}

__global__ __launch_bounds__(256) void k_lc_sqsum(const float* g, int n, float* part) {
    __shared__ float s_red[4];
    float s = 0.0f;
    const int i0 = blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = i0 + k * 256;
        if (i < n) s = fmaf(g[i], g[i], s);
    }
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
}
struct LcAdam {
    float lr, beta1, beta2, eps, weight_decay, max_norm, bc1, bc2;
    int sq_blocks, n;
};
// torch.optim.Adam with L2 weight decay in the gradient (gomoku/run_training.py builds it so), clip_grad_norm_ coefficient from the partials
__global__ __launch_bounds__(256) void k_lc_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                  const float* __restrict__ sq_part, const LcAdam a) {
    __shared__ float s_coef;
    if (threadIdx.x == 0) {
        float coef = 1.0f;
        if (a.max_norm > 0.0f) {
            double s = 0.0;
            for (int b = 0; b < a.sq_blocks; b++) s += (double)sq_part[b];
            const float c = a.max_norm / ((float)sqrt(s) + 1e-6f);
            coef = c < 1.0f ? c : 1.0f;
        }
        s_coef = coef;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    float gi = g[i] * s_coef;
    const float pi = p[i];
    gi = gi + a.weight_decay * pi;
    const float mi = a.beta1 * m[i] + (1.0f - a.beta1) * gi;
    const float vi = a.beta2 * v[i] + (1.0f - a.beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(a.bc2) + a.eps;
    p[i] = pi - (a.lr / a.bc1) * (mi / denom);
}

}  // namespace mzlc
