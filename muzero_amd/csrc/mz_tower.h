// mz_tower.h -- a whole residual tower (network.py:293-299 repeated R times: conv-BN-ReLU-conv-BN, += x, ReLU) as ONE
// persistent kernel for small boards, activations resident in LDS.
//
// For small hidden states (Atari's 6x6, TicTacToe's 3x3, boards up to ~8x8) one k_conv3x3 launch is only ~50 us of work and a
// fifth of it is fixed cost (prologue, first-fetch latency, epilogue drain, launch ramp), paid 2R times per tower.  Here a
// workgroup owns G whole images and ALL output channels, so consecutive convolutions depend only on the workgroup's own data:
// no grid-wide dependency, no HBM round trip between layers.
//   workgroup = 512 threads = 8 waves, wave w = output-channel tile w (16 channels; P <= 128), NPT pixel tiles of 16 over the
//   G * h * w pixels of the group (accumulators in registers, D[pixel][co] as in k_conv3x3: same fmaf-chain order
//   (16-channel block, tap, channel), same bias start, same residual-then-ReLU epilogue => bit-identical results).
//   LDS holds three activation buffers (x, conv1 output, block output; rotating) in the A-operand layout of k_conv3x3's slab
//   without a halo: act[cb][slot q][position][4] = channels {q, 4+q, 8+q, 12+q} of block cb; a tap that falls outside the
//   image reads a 16-byte zero slot instead (fma(0, w, acc) == acc).  Weights stream from L2 through an 8-step register ring
//   per wave, straight across layer boundaries.  One barrier per convolution.
#pragma once
#include "mz_conv.h"

namespace mz {

struct TowerLaunch {
    const float* in;            // dense [B][P][hw]
    float* out;                 // dense [B][P][hw]
    const float* const* w;      // [n_convs] packed conv weights (k_conv3x3 layout: [co_tile][cb][tap][64][4])
    const float* const* bias;   // [n_convs] padded biases
    int n_convs;                // 2 * blocks: even index = conv1 (ReLU), odd = conv2 (+ block input, ReLU)
    int P, h, w_img, G, B;
    int nposp;                  // positions per plane, multiple of 16 (>= G * h * w)
    long long* stamps;
};

template <int NPT>
__global__ __launch_bounds__(512, 1) void k_res_tower(const TowerLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
    const int hw = L.h * L.w_img, npix = L.G * hw, img0 = blockIdx.x * L.G;
    const int n_cb = L.P >> 4, co_tiles = n_cb;
    const int plane = L.nposp * 4;                 // floats per (cb, slot) plane
    const int bufsz = n_cb * 4 * plane;            // floats per activation buffer
    const int zero_off = 3 * bufsz;                // 16-byte zero slot after the three buffers
    const bool active = wave < co_tiles;           // waves beyond the channel tiles only take part in barriers
    const int cot = active ? wave : 0;
    const float r_hw = 1.0f / (float)hw, r_w = 1.0f / (float)L.w_img;

    // ---- load the group's input into buffer 0 (layout transform), zero the padding positions and the zero slot ----
    for (int i = tid; i < 3 * bufsz / 4 + 1; i += 512) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (int i = tid; i < L.P * npix; i += 512) {  // i = (g, c, p): consecutive threads read consecutive floats of the dense input
        const int g = i / (L.P * hw), r = i - g * L.P * hw, c = r / hw, p = r - c * hw;
        if (img0 + g < L.B) {
            const float v = L.in[((size_t)(img0 + g) * L.P + c) * hw + p];
            lds[(((c >> 4) * 4 + (c & 3)) * L.nposp + g * hw + p) * 4 + ((c & 15) >> 2)] = v;
        }
    }
    // ---- per-lane pixel bookkeeping: A-operand rows are pixel slots pt*16 + j; accumulator rows are slots pt*16 + 4q + r ----
    int pos4[NPT];       // float offset (position * 4 + q * plane) of this lane's A row at tap (0,0) relative to block 0
    unsigned vmask[NPT]; // bit t set: tap t of this lane's pixel lies inside the image
#pragma unroll
    for (int pt = 0; pt < NPT; pt++) {
        const int p = pt * 16 + j, g = conv_idiv(p, r_hw), pp = p - g * hw, y = conv_idiv(pp, r_w), x = pp - y * L.w_img;
        pos4[pt] = p * 4 + q * plane;
        unsigned m = 0;
        if (p < npix) {
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (yy >= 0 && yy < L.h && xx >= 0 && xx < L.w_img) m |= 1u << t;
            }
        }
        vmask[pt] = m;
    }
    const int n_steps = n_cb * 9;
    constexpr int WD = 9;  // weight ring: step s of the current conv in wr[s % 9], 8 steps ahead (9 | 9: static ring index)
    float4 wr[WD];
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.w[0]), 0, -1, 0x00020000);
    const int wbase = cot * n_steps * 1024;
    auto wload = [&](const __amdgpu_buffer_rsrc_t& rs, int step) {
        const conv_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, wbase + step * 1024, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
#pragma unroll
    for (int s0 = 0; s0 < WD - 1; s0++) wr[s0] = wload(rs_w, s0);
    __syncthreads();

    int b_in = 0, b_t1 = 1, b_out = 2;  // buffer roles: block input x, conv1 output, block output
    for (int cv = 0; cv < L.n_convs; cv++) {
        const bool second = cv & 1;
        const float* src = lds + (second ? b_t1 : b_in) * bufsz;
        float* dst = lds + (second ? b_out : b_t1) * bufsz;
        const float* res = lds + b_in * bufsz;
        // next conv's weight descriptor: the ring crosses the layer boundary (steps >= n_steps come from conv cv + 1)
        const int cvn = cv + 1 < L.n_convs ? cv + 1 : cv;
        const __amdgpu_buffer_rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.w[cvn]), 0, -1, 0x00020000);
        f32x4 acc[NPT];
        {
            const float bv = L.bias[cv][cot * 16 + j];
#pragma unroll
            for (int pt = 0; pt < NPT; pt++) acc[pt] = f32x4{bv, bv, bv, bv};
        }
        auto a_read = [&](int cb, int tap, int pt) {  // A operand: 4 channels of block cb at the tap-shifted pixel, or zeros
            const int d = ((tap / 3 - 1) * L.w_img + (tap % 3 - 1)) * 4;
            const int off = ((vmask[pt] >> tap) & 1u) ? cb * 4 * plane + pos4[pt] + d : zero_off;
            return *reinterpret_cast<const float4*>(src + off);
        };
        float4 xr[3];
        for (int cb = 0; cb < n_cb; cb++) {
            xr[0] = a_read(cb, 0, 0);
            xr[1] = NPT > 1 ? a_read(cb, 0, NPT > 1 ? 1 : 0) : a_read(cb, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                {   // weights 8 steps ahead; past this conv's last step the stream continues with the next conv's first steps
                    const int st = cb * 9 + tap + WD - 1;
                    wr[(tap + WD - 1) % WD] = st < n_steps ? wload(rs_w, st) : wload(rs_n, st - n_steps < n_steps ? st - n_steps : n_steps - 1);
                }
#pragma unroll
                for (int pt = 0; pt < NPT; pt++) {
                    const int n = tap * NPT + pt, n2 = n + 2;
                    if (n2 < 9 * NPT) {
                        const int tap2 = n2 / NPT, pt2 = n2 - tap2 * NPT;
                        xr[n2 % 3] = a_read(cb, tap2, pt2);
                    }
                    const float4 x4 = xr[n % 3], w4 = wr[tap % WD];
                    acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.x, w4.x, acc[pt], 0, 0, 0);
                    acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.y, w4.y, acc[pt], 0, 0, 0);
                    acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.z, w4.z, acc[pt], 0, 0, 0);
                    acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.w, w4.w, acc[pt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
                for (int pt = 0; pt < NPT; pt++) {
                    if (tap * NPT + pt + 2 < 9 * NPT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue: lane (q, j) holds pixel slots pt*16 + 4q + r of channel 16*cot + j: (+ block input), ReLU, store ----
        const bool last = cv + 1 == L.n_convs;
        if (active) {
            const int co = cot * 16 + j;
            const int dbase = ((cot * 4 + (j & 3)) * L.nposp) * 4 + (j >> 2);  // channel co inside a buffer
#pragma unroll
            for (int pt = 0; pt < NPT; pt++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int p = pt * 16 + 4 * q + r;
                    if (p < npix) {
                        float v = acc[pt][r];
                        v = v + (second ? res[dbase + p * 4] : 0.0f);
                        if (!(v > 0.0f)) v = 0.0f;
                        if (last) {
                            const int g = conv_idiv(p, r_hw), pp = p - g * hw;
                            if (img0 + g < L.B) L.out[((size_t)(img0 + g) * L.P + co) * hw + pp] = v;
                        } else {
                            dst[dbase + p * 4] = v;
                        }
                    }
                }
            }
        }
        if (second) { const int t = b_in; b_in = b_out; b_out = t; }
        rs_w = rs_n;
        __syncthreads();
    }
}

}  // namespace mz
