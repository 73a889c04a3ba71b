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
//   image reads a padding position of the plane, which stays zero (fma(0, w, acc) == acc).  Weights stream from L2 through an 8-step register ring
//   per wave, straight across layer boundaries.  One barrier per convolution.
#pragma once
#include "mz_conv.h"

namespace mz {

struct TowerLaunch {
    const float* in;            // dense [B][P][hw]
    float* out;                 // dense [B][P][hw]
    const float* w;             // [n_convs] packed conv weights back to back (each in k_conv3x3's layout [co_tile][cb][tap][64][4]):
                                // one buffer descriptor, the weight stream of a wave is linear across layer boundaries
    const float* bias;          // [n_convs][P] folded biases
    int n_convs;                // 2 * blocks: even index = conv1 (ReLU), odd = conv2 (+ block input, ReLU)
    int P, h, w_img, G, B;
    int nposp;                  // positions per plane, multiple of 16, > G * h * w (at least one zero padding position)
    long long* stamps;
};

// SPEC == 1: the Atari nets' tower -- 128 planes, two 6 x 6 images per workgroup -- with those numbers as compile-time constants (the
// launcher checks them): C4 +0.4 %.  (Positions per plane, `nposp`, stays a kernel argument: as a constant too it costs 2.5 % --
// this kernel's operand offsets already live in VGPRs, and the constant plane stride only changed the schedule for the worse.)
template <int NPT, int SPEC = 0>
__global__ __launch_bounds__(512, 1) void k_res_tower(const TowerLaunch L_) {
    TowerLaunch L = L_;
    if constexpr (SPEC == 1) { L.P = 128; L.h = 6; L.w_img = 6; L.G = 2; }
    MZC_T_DECL
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
    const int hw = L.h * L.w_img, npix = L.G * hw, img0 = blockIdx.x * L.G;
    const int n_cb = L.P >> 4, co_tiles = n_cb;
    const int plane = L.nposp * 4;                 // floats per (cb, slot) plane
    const int bufsz = n_cb * 4 * plane;            // floats per activation buffer
    const bool active = wave < co_tiles;           // waves beyond the channel tiles only take part in barriers
    const int cot = active ? wave : 0;
    const float r_hw = 1.0f / (float)hw, r_w = 1.0f / (float)L.w_img;

    // ---- load the group's input into buffer 0 (layout transform), zero the padding positions and the zero slot ----
    for (int i = tid; i < 3 * bufsz / 4; i += 512) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (int i = tid; i < L.P * npix; i += 512) {  // i = (g, c, p): consecutive threads read consecutive floats of the dense input
        const int g = i / (L.P * hw), r = i - g * L.P * hw, c = r / hw, p = r - c * hw;
        if (img0 + g < L.B) {
            const float v = L.in[((size_t)(img0 + g) * L.P + c) * hw + p];
            lds[(((c >> 4) * 4 + (c & 3)) * L.nposp + g * hw + p) * 4 + ((c & 15) >> 2)] = v;
        }
    }
    // ---- per-lane pixel bookkeeping: A-operand rows are pixel slots pt*16 + j; accumulator rows are slots pt*16 + 4q + r.
    // aoff[tap][pt]: float offset, inside channel block 0, of the lane's A operand for that tap -- the tap-shifted pixel, or
    // the first padding position of the plane (always zero: nposp > G * h * w) when the tap falls outside the image.  All
    // 9 * NPT offsets live in VGPRs so that the inner loop spends ONE VALU instruction per LDS read (VALU instructions
    // take MFMA issue slots: with 5 per read the tap loop ran 25 % slower) ----
    int aoff[9][NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; pt++) {
        const int p = pt * 16 + j, g = conv_idiv(p, r_hw), pp = p - g * hw, y = conv_idiv(pp, r_w), x = pp - y * L.w_img;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            const bool ok = p < npix && yy >= 0 && yy < L.h && xx >= 0 && xx < L.w_img;
            aoff[t][pt] = q * plane + (ok ? p + (t / 3 - 1) * L.w_img + (t % 3 - 1) : npix) * 4;
        }
    }
    // only the last pixel tile can hold slots beyond the group's pixels (slots are contiguous); their outputs are forced to 0,
    // which keeps the padding positions of every activation buffer zero
    float tail_keep[4];
#pragma unroll
    for (int r = 0; r < 4; r++) tail_keep[r] = ((NPT - 1) * 16 + 4 * q + r < npix) ? 1.0f : 0.0f;
    const int n_steps = n_cb * 9;
    constexpr int WD = 9;  // weight ring: step s of the current conv in wr[s % 9], 8 steps ahead (9 | 9: static ring index)
    float4 wr[WD];
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.w), 0, -1, 0x00020000);
    const int conv_bytes = co_tiles * n_steps * 1024;  // one conv's packed weights
    const int wbase = cot * n_steps * 1024;            // this wave's channel tile inside a conv
    auto wload = [&](int cv, int step) {               // step may run past the conv's end into the next conv's first steps
        const int over = step >= n_steps ? 1 : 0;
        const int cvx = cv + over < L.n_convs ? cv + over : cv;
        const conv_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, cvx * conv_bytes + wbase + (step - over * n_steps) * 1024, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
#pragma unroll
    for (int s0 = 0; s0 < WD - 1; s0++) wr[s0] = wload(0, s0);
    __syncthreads();

    MZC_T(0);
    int b_in = 0, b_t1 = 1, b_out = 2;  // buffer roles: block input x, conv1 output, block output
    for (int cv = 0; cv < L.n_convs; cv++) {
        const bool second = cv & 1;
        const float* src = lds + (second ? b_t1 : b_in) * bufsz;
        float* dst = lds + (second ? b_out : b_t1) * bufsz;
        const float* res = lds + b_in * bufsz;
        f32x4 acc[NPT];
        {
            const float bv = L.bias[cv * L.P + cot * 16 + j];
#pragma unroll
            for (int pt = 0; pt < NPT; pt++) acc[pt] = f32x4{bv, bv, bv, bv};
        }
        auto a_read = [&](int cb, int tap, int pt) {  // A operand: 4 channels of block cb at the tap-shifted pixel (or zeros)
            return *reinterpret_cast<const float4*>(src + cb * 4 * plane + aoff[tap][pt]);
        };
        float4 xr[3];
        // the A-operand ring (step n = tap * NPT + pt of a block in xr[n % 3], two steps ahead) runs across block boundaries:
        // 9 * NPT is a multiple of 3, so the slot of a step does not depend on the block
        xr[0] = a_read(0, 0, 0);
        xr[1] = NPT > 1 ? a_read(0, 0, NPT > 1 ? 1 : 0) : a_read(0, 1, 0);
        MZC_T(1);
        for (int cb = 0; cb < n_cb; cb++) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                wr[(tap + WD - 1) % WD] = wload(cv, cb * 9 + tap + WD - 1);  // weights 8 steps ahead, across the layer boundary
#pragma unroll
                for (int pt = 0; pt < NPT; pt++) {
                    const int n = tap * NPT + pt, n2 = n + 2;
                    if (n2 < 9 * NPT) {
                        const int tap2 = n2 / NPT, pt2 = n2 - tap2 * NPT;
                        xr[n2 % 3] = a_read(cb, tap2, pt2);
                    } else {  // first steps of the next block (the last block re-reads its own: harmless, never consumed)
                        const int n3 = n2 - 9 * NPT, tap3 = n3 / NPT, pt3 = n3 - tap3 * NPT;
                        xr[n2 % 3] = a_read(cb + 1 < n_cb ? cb + 1 : cb, tap3, pt3);
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
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            MZC_T(2);
        }
        // ---- epilogue: lane (q, j) holds pixel slots pt*16 + 4q + r of channel 16*cot + j: (+ block input), ReLU, store ----
        const bool last = cv + 1 == L.n_convs;
        if (active) {
            const int co = cot * 16 + j;
            const int dbase = ((cot * 4 + (j & 3)) * L.nposp) * 4 + (j >> 2);  // channel co inside a buffer
            f32x4 rv[NPT];  // block input at this lane's outputs: all reads first, then the stores (the compiler cannot reorder them
                            // itself: both go to the same LDS array); padding positions read 0
#pragma unroll
            for (int pt = 0; pt < NPT; pt++) {
#pragma unroll
                for (int r = 0; r < 4; r++) rv[pt][r] = second ? res[dbase + (pt * 16 + 4 * q + r) * 4] : 0.0f;
            }
#pragma unroll
            for (int pt = 0; pt < NPT; pt++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int p = pt * 16 + 4 * q + r;
                    float v = acc[pt][r] + rv[pt][r];
                    if (!(v > 0.0f)) v = 0.0f;
                    if (pt == NPT - 1) v = v * tail_keep[r];
                    if (last) {
                        const int g = conv_idiv(p, r_hw), pp = p - g * hw;
                        if (p < npix && img0 + g < L.B) L.out[((size_t)(img0 + g) * L.P + co) * hw + pp] = v;
                    } else {
                        dst[dbase + p * 4] = v;
                    }
                }
            }
        }
        if (second) { const int t = b_in; b_in = b_out; b_out = t; }
        MZC_T(3);
        __syncthreads();
        MZC_T(4);
    }
    MZC_T_FLUSH(L);
}

}  // namespace mz
