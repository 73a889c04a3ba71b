// mz_conv.h -- conv-tower inference kernels (MuZeroBoardGameNet network.py:540-574, MuZeroAtariNet :501-537) for gfx950.
//
// k_conv3x3<NPT, NCT, WHOLE>: 3x3 convolution (stride 1 or 2, pad 1) with eval-mode BatchNorm folded into weight/bias,
// optional residual add and ReLU (ResNetBlock network.py:293-299), as an implicit GEMM on v_mfma_f32_16x16x4_f32:
//     D[pixel][co] = bias[co] + sum_k X[pixel][k] * W[co][k],   k = (16-channel block, tap (ky,kx), channel in block)
// B operand = 4 k x 16 output channels of the weights (pre-packed fragment order, a linear 1 KiB-per-step stream per wave
// out of L2), A operand = 16 output pixels x 4 channels read from an LDS-staged input slab (tile + halo, one 16-byte read
// per lane and (tap, pixel tile)).  The k order is ONE fmaf chain per output in exactly the oracle's order, so results
// equal the oracle bit for bit; zero padding contributes fma(0, w, acc) == acc.
//   workgroup = 256 threads = a group of G images (blockIdx.y) x one spatial tile (blockIdx.x) = up to NPT*16 output
//   pixel slots x one slice of 64*NCT output channels (blockIdx.z; wave w owns channel tiles w, w+4 of the slice).
//   Small boards pack several images into the pixel dimension (6x6: 4 images = 144 pixels = 9 full MFMA tiles).
//   Pipeline per 16-channel block: the NEXT block's slab is fetched global -> registers between the MFMAs of this block
//   (a few buffer loads per tap) and written to the other LDS buffer after the tap loop (one barrier per block); weights
//   run 2 or 8 steps ahead in a register ring; A operands are read two pixel tiles ahead of the MFMAs that consume them;
//   sched_group_barrier pins that order.  All index arithmetic is hoisted out of the loop.
//   WHOLE = the tile is the whole image (stride 1; the shapes of the search): lanes stage pixel quads with 16-byte loads.
// The dynamics net's action planes (network.py:440-444: element f = c*h*w + y*w + x of the [A,h,w] block is 1 iff
// f % A == action) are generated while staging, never materialised.
#pragma once
#include "mz_mlp.h"

namespace mz {

struct ConvLaunch {
    // input: per-image base pointers (gather from the node store) or a dense buffer
    const float* const* in_ptrs;  // [B] or null
    const float* in;              // dense [B][cin_real][ih][iw] if in_ptrs == null
    const float* in_base;         // with in_ptrs and G > 1: the node store's base; every in_ptrs[b] - in_base must be < 1 Gi floats
    const int* action;            // [B] or null: channels >= cin_real are action planes over num_actions
    int num_actions;
    int cin_real;                 // channels present in memory
    int cin;                      // logical input channels (cin_real + num_actions planes), k runs over pad16(cin)
    int ih, iw, oh, ow, stride;
    int cout;                     // output channels; blockIdx.z selects a slice of 64*NCT
    const float* w;               // packed [co_tile][cb][tap][64 lanes][4]
    const float* bias;            // [pad16(cout)]
    const float* residual;        // dense [B][cout][oh][ow] or null
    float* out;                   // dense [B][cout][oh][ow]
    int relu;
    int th, tw;                   // spatial tile of one image (G * th * tw <= NPT*16)
    int tiles_x, tiles_y;
    int G;                        // images per workgroup (> 1 only when one tile covers the whole image)
    int cstride;                  // LDS floats per slab buffer = 4 * qstride
    int qstride;                  // LDS floats per channel-slot plane (>= 4 * G * sih * siw, multiple of 64: see the slab layout)
    int B;
    long long* stamps;            // diagnostic builds (-DMZC_STAMPS) only
    // SP builds (see k_action_sparse below): the <= 9 non-zero action-plane terms of the dynamics net's first conv, added in the epilogue
    const float* sp_w;            // [A * 9 + 1][cout]: row c * 9 + tap = folded weights of action channel c, last row zeros
    const int* sp_terms;          // [A][ih * iw][12]: byte offsets of the pixel's nine rows of sp_w in chain order (absent terms: the zero row)
    const int* sp_action;         // [B]
};

constexpr int CONV_RK = 2;    // slab positions per thread: G * sih * siw <= 384

typedef unsigned int conv_u32x4 __attribute__((ext_vector_type(4)));

// floor(p / d) for 0 <= p < 4096 via a float reciprocal (index arithmetic only): (p + 0.5) / d is at least 0.5 / d away from
// an integer, far more than float rounding, so the truncation is exact
__device__ __forceinline__ int conv_idiv(int p, float rcp_d) { return (int)(((float)p + 0.5f) * rcp_d); }

#ifdef MZC_STAMPS  // per-phase cycle sums of wave 0 of workgroup (0,0,0): [0] prologue [1] first A reads of the block [2] tap loop [3] store [4] barrier [5] epilogue
#define MZC_T_DECL long long _ct0 = __builtin_readcyclecounter(), _cacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MZC_T(i) do { const long long _t = __builtin_readcyclecounter(); _cacc[i] += _t - _ct0; _ct0 = _t; } while (0)
#define MZC_T_FLUSH(L) do { if (L.stamps && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) for (int _i = 0; _i < 8; _i++) L.stamps[_i] = _cacc[_i]; } while (0)
#else
#define MZC_T_DECL
#define MZC_T(i) do {} while (0)
#define MZC_T_FLUSH(L) do {} while (0)
#endif

#ifdef MZC_NO_XS  // diagnostic variants (tools/micro/conv_bench.hip), never defined in the product build
#define MZC_XS_READ 0
#else
#define MZC_XS_READ 1
#endif

// All global reads of the main loop are buffer loads: address = descriptor base (SGPRs, workgroup- or wave-uniform) +
// per-lane byte offset (ONE VGPR, loop-invariant) + uniform byte offset (SGPR: channel or weight step).  With plain
// pointers hipcc keeps one 64-bit VGPR address per unrolled load alive across the loop (or emits flat loads that also
// tick lgkmcnt and serialise against the LDS reads).
template <int NPT, int NCT, bool WHOLE, int SIDE = 0, bool SP = false>
__global__ __launch_bounds__(256, (NCT * NPT > 15) ? 1 : 2) void k_conv3x3(const ConvLaunch L_) {
    static_assert(!SP || WHOLE, "the sparse action terms are fused into whole-image builds only");
    // SIDE > 0: the launch geometry of a whole SIDE x SIDE image per workgroup, 128 output channels, as compile-time constants (the
    // launcher checks every one).  The kernel's index arithmetic is hoisted out of its loops, but with the geometry in kernel
    // arguments it is still ~4 % of a 140 us launch (prologue divisions, per-tap offsets, predicates): C5 0.757 -> 0.785 of the peak.
    ConvLaunch L = L_;
    if constexpr (SIDE > 0 && WHOLE) {
        L.ih = SIDE; L.iw = SIDE; L.oh = SIDE; L.ow = SIDE; L.stride = 1; L.th = SIDE; L.tw = SIDE; L.tiles_x = 1; L.tiles_y = 1; L.G = 1; L.cout = 128;
        L.qstride = ((SIDE + 2) * (SIDE + 2) * 4 + 63) & ~63; L.cstride = 4 * L.qstride;
    }
    if constexpr (SIDE > 0 && !WHOLE) {  // tiled: a SIDE x SIDE image in 12 x (NPT == 12 ? 16 : 12) output tiles, stride 1, 128 output channels
        constexpr int TW = NPT == 12 ? 16 : 12;
        L.ih = SIDE; L.iw = SIDE; L.oh = SIDE; L.ow = SIDE; L.stride = 1; L.th = 12; L.tw = TW; L.tiles_x = SIDE / TW; L.tiles_y = SIDE / 12; L.G = 1; L.cout = 128;
        L.qstride = (14 * (TW + 2) * 4 + 63) & ~63; L.cstride = 4 * L.qstride;
    }
    MZC_T_DECL
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // slab[2][4 slots][qstride]: position r = g * plane + sy * siw + sx keeps its 16 channels as four float4s, slot q at
    // q * qstride + 4r = {ch q, ch 4+q, ch 8+q, ch 12+q}: the A operands of the four k-steps of lane group q.  The 16 lanes
    // of a group read consecutive positions = 64 consecutive dwords, and qstride is a multiple of 64 dwords, which makes
    // the 16-byte reads of ds_read_b128's mixed-q lane groups conflict-free (PMC: 65 % of LDS cycles were conflicts with a
    // position-major [r][16 + 4] layout)
    float* slab = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
    const int tile = blockIdx.x, ty0 = (tile / L.tiles_x) * L.th, tx0 = (tile % L.tiles_x) * L.tw;
    const int sih = (L.th - 1) * L.stride + 3, siw = (L.tw - 1) * L.stride + 3, plane = sih * siw;
    const int img0 = blockIdx.y * L.G, TP = L.th * L.tw, ihw = L.ih * L.iw;
    const int co_tiles = (L.cout + 15) >> 4, n_cb = (L.cin + 15) >> 4, bufsz = L.cstride;  // floats per slab buffer
    const int iy0 = ty0 * L.stride - 1, ix0 = tx0 * L.stride - 1;
    const float r_plane = 1.0f / (float)plane, r_siw = 1.0f / (float)siw, r_tp = 1.0f / (float)TP, r_tw = 1.0f / (float)L.tw;

    const int img0c = img0 < L.B ? img0 : L.B - 1;
    const float* ibase = L.in_ptrs ? (L.G == 1 ? L.in_ptrs[img0c] : L.in_base) : L.in;  // workgroup-uniform
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ibase), 0, -1, 0x00020000);
    auto pure_real = [&](int cb) { return cb * 16 + 16 <= L.cin_real; };
    // ---- staging plan, tile == whole image (stride 1): a channel of an image is hw contiguous floats.  Lane t of every wave
    // owns pixel quad t of the group (4 consecutive pixels, one 16-byte load per channel); wave w owns the channels
    // {w, 4+w, 8+w, 12+w} of each 16-channel block, i.e. exactly the float4 at slot 4w of each slab position: per block and
    // thread 4 loads and 4 16-byte LDS writes.  The halo is zeroed once; it is never written again. ----
    const int QP = (ihw + 3) >> 2;  // quads per image; the host guarantees G * QP <= 64
    unsigned w_voff = 0;
    int w_spos[4], w_pm[4], w_act = -1;
    bool w_ok = false;
    f32x4 w_sv[4];  // [i]: channel 4i + wave of the block, pixels p0 .. p0+3
    if constexpr (WHOLE) {
        const float r_qp = 1.0f / (float)QP, r_iw = 1.0f / (float)L.iw;
        const int g = conv_idiv(lane, r_qp), qd = lane - g * QP, p0 = qd * 4, bimg = img0 + g;
        w_ok = lane < L.G * QP && bimg < L.B;
        const int cimg = bimg < L.B ? bimg : L.B - 1;
        size_t o = (size_t)(w_ok ? p0 : 0);
        if (!L.in_ptrs) o += (size_t)cimg * L.cin_real * ihw;
        else if (L.G != 1) o += (size_t)(L.in_ptrs[cimg] - L.in_base);
        w_voff = (unsigned)(o * sizeof(float));
        w_act = (w_ok && L.action) ? L.action[cimg] : -1;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int pp = p0 + e, py = conv_idiv(pp, r_iw), px = pp - py * L.iw;
            w_spos[e] = (w_ok && pp < ihw) ? ((g < L.G ? g : 0) * plane + (py + 1) * siw + px + 1) * 4 + wave * L.qstride : -1;
            w_pm[e] = L.cin > L.cin_real ? pp % L.num_actions : 0;
        }
        for (int i = tid; i < bufsz / 2; i += 256) reinterpret_cast<float4*>(slab)[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // both buffers
        if constexpr (SP) {
            // this workgroup's term rows (12 ints per pixel, per image its action's table) -> LDS behind the slabs; read in the epilogue,
            // many barriers from here
            int4* s_term4 = reinterpret_cast<int4*>(slab + 2 * bufsz);
            const float r_row = 1.0f / (float)(ihw * 3);
            for (int i = tid; i < L.G * ihw * 3; i += 256) {
                const int g = conv_idiv(i, r_row), rem = i - g * ihw * 3, cimg = img0 + g < L.B ? img0 + g : L.B - 1;
                s_term4[i] = reinterpret_cast<const int4*>(L.sp_terms)[(size_t)L.sp_action[cimg] * ihw * 3 + rem];
            }
        }
        __syncthreads();
    }
    auto wfetch_real = [&](int cb, int i) {  // channel 4i + wave of block cb, clamped to a valid channel (branch-free)
#ifndef MZC_NO_FETCH
        const int ch = cb * 16 + 4 * i + wave, chc = ch < L.cin_real ? ch : 0;
        const conv_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_in, w_voff, chc * ihw * (int)sizeof(float), 0);
        w_sv[i] = f32x4{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
#else
        w_sv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
    };
    auto wfix_generic = [&](int cb) {  // overwrite the lanes of action-plane (network.py:440-444) and padding channels of block cb
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int ch = cb * 16 + 4 * i + wave;  // wave-uniform
            if (ch >= L.cin_real) {
                const int t = ch < L.cin ? (int)(((long long)(ch - L.cin_real) * ihw) % L.num_actions) : 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    int m = w_pm[e] + t;
                    m = m >= L.num_actions ? m - L.num_actions : m;
                    w_sv[i][e] = (ch < L.cin && m == w_act) ? 1.0f : 0.0f;
                }
            }
        }
    };
    auto wstore = [&](int buf) {
        float* d = slab + buf * bufsz;
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (w_spos[e] >= 0) *reinterpret_cast<float4*>(d + w_spos[e]) = make_float4(w_sv[0][e], w_sv[1][e], w_sv[2][e], w_sv[3][e]);
    };
    // ---- staging plan, tiled images (any stride): positions r = tid, tid + 256; out-of-image positions read a clamped address and are zeroed ----
    unsigned voff[CONV_RK];
    int am[CONV_RK], sact[CONV_RK];
    bool swrite[CONV_RK], sval[CONV_RK];
    const int ihwA = ihw % L.num_actions;
#pragma unroll
    for (int k = 0; k < CONV_RK; k++) {
        const int r = tid + 256 * k;
        swrite[k] = !WHOLE && r < L.G * plane;
        const int rc = swrite[k] ? r : 0;
        const int g = conv_idiv(rc, r_plane), rr = rc - g * plane, sy = conv_idiv(rr, r_siw), sx = rr - sy * siw;
        const int gy = iy0 + sy, gx = ix0 + sx, bimg = img0 + g;
        sval[k] = swrite[k] && bimg < L.B && gy >= 0 && gy < L.ih && gx >= 0 && gx < L.iw;
        const int cy = gy < 0 ? 0 : (gy >= L.ih ? L.ih - 1 : gy), cx = gx < 0 ? 0 : (gx >= L.iw ? L.iw - 1 : gx);
        const int cimg = bimg < L.B ? bimg : L.B - 1;
        size_t o = (size_t)(cy * L.iw + cx);
        if (!L.in_ptrs) o += (size_t)cimg * L.cin_real * ihw;
        else if (L.G != 1) o += (size_t)(L.in_ptrs[cimg] - L.in_base);
        voff[k] = (unsigned)(o * sizeof(float));
        am[k] = L.cin > L.cin_real ? (cy * L.iw + cx) % L.num_actions : 0;  // flat index of the first action channel at this pixel, mod A
        sact[k] = (sval[k] && L.action) ? L.action[cimg] : -1;
    }
    float sv[CONV_RK][16];
    // global -> registers for channel block cb.  A block of 16 real channels is a straight run of 16 * CONV_RK independent
    // buffer loads (per-channel branches make hipcc wait for every load before the next); inside the main loop those loads
    // are issued a few per tap between the MFMAs (fetch_part), because the memory pipeline accepts them slowly and a wave
    // that issues all of them at once leaves its MFMA pipe idle meanwhile.  Blocks holding action planes / padding channels
    // take the generic path.
    auto fetch_part = [&](int cb, int c_lo, int c_hi) {  // channels [c_lo, c_hi) of block cb, clamped to a valid channel
#ifndef MZC_NO_FETCH
#pragma unroll
        for (int c = c_lo; c < c_hi; c++) {
            const int ch = cb * 16 + c, chc = ch < L.cin_real ? ch : 0;  // workgroup-uniform
#pragma unroll
            for (int k = 0; k < CONV_RK; k++)  // raw value from the clamped address; out-of-image positions are zeroed in store()
                sv[k][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in, voff[k], chc * ihw * (int)sizeof(float), 0));
        }
#endif
    };
    auto fetch_generic = [&](int cb) {  // blocks with action planes (network.py:440-444) and / or the zero channels padding cin to 16
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int ch = cb * 16 + c;
#pragma unroll
            for (int k = 0; k < CONV_RK; k++) {
                float v = 0.0f;
                if (ch < L.cin_real) {
#ifndef MZC_NO_FETCH
                    v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in, voff[k], ch * ihw * (int)sizeof(float), 0));
#endif
                } else if (ch < L.cin) {
                    v = (am[k] == sact[k]) ? 1.0f : 0.0f;
                    am[k] += ihwA;
                    am[k] = am[k] >= L.num_actions ? am[k] - L.num_actions : am[k];
                }
                sv[k][c] = v;
            }
        }
    };
    auto store = [&](int buf, int cb) {
        float* d = slab + buf * bufsz;
#pragma unroll
        for (int k = 0; k < CONV_RK; k++)
            if (swrite[k]) {
#pragma unroll
                for (int c = 0; c < 16; c++)
                    if (cb * 16 + c < L.cin_real && !sval[k]) sv[k][c] = 0.0f;  // zero padding of the real channels
                float* o = d + (tid + 256 * k) * 4;
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++)
                    *reinterpret_cast<float4*>(o + g4 * L.qstride) = make_float4(sv[k][g4], sv[k][4 + g4], sv[k][8 + g4], sv[k][12 + g4]);
            }
    };

    // ---- A-operand rows of this lane: pixel slot p = pt*16 + j -> image g of the group, (py, px) inside the tile ----
    int off[NPT];   // float offset of the lane's float4 for tap (0,0)
#pragma unroll
    for (int pt = 0; pt < NPT; pt++) {
        const int p = pt * 16 + j, g = conv_idiv(p, r_tp), pp = p - g * TP, py = conv_idiv(pp, r_tw), px = pp - py * L.tw;
        off[pt] = (g < L.G ? g * plane + (py * L.stride) * siw + px * L.stride : 0) * 4 + q * L.qstride;
    }
    // ---- accumulators D[pixel slot 4q + r][channel j] start at the bias; the weight stream of channel tile c is linear in
    // (cb, tap): 1 KiB per step ----
    f32x4 acc[NCT][NPT];
    int cot[NCT], wbase[NCT];
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.w), 0, -1, 0x00020000);
#pragma unroll
    for (int c = 0; c < NCT; c++) {
        cot[c] = blockIdx.z * 4 * NCT + wave + 4 * c;
        const int ct = cot[c] < co_tiles ? cot[c] : co_tiles - 1;  // out-of-range tiles compute a duplicate that is never stored
        const float bv = L.bias[ct * 16 + j];
#pragma unroll
        for (int pt = 0; pt < NPT; pt++) acc[c][pt] = f32x4{bv, bv, bv, bv};
        wbase[c] = ct * n_cb * 9 * 1024;  // bytes; wave-uniform
    }
    const int n_steps = n_cb * 9;  // weight steps (cb, tap)
    auto wload = [&](int c, int step) {
        const int sc = step < n_steps ? step : n_steps - 1;
        const conv_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, wbase[c] + sc * 1024, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    // weight ring: step s lives in wr[s % WD], WD - 1 steps ahead of its use.  An L2 hit takes ~1 us: short taps (few pixel
    // tiles) need the deep ring; 9 % WD == 0 keeps the ring index static under the unrolled tap loop
    constexpr int WD = (NPT * NCT <= 9) ? 9 : 3;
    float4 wr[WD][NCT];
#pragma unroll
    for (int c = 0; c < NCT; c++) {
#pragma unroll
        for (int s0 = 0; s0 < WD - 1; s0++) wr[s0][c] = wload(c, s0);
    }
    if constexpr (WHOLE) {
#pragma unroll
        for (int i = 0; i < 4; i++) wfetch_real(0, i);
        if (!pure_real(0)) wfix_generic(0);
        wstore(0);
    } else {
        if (pure_real(0)) fetch_part(0, 0, 16);
        else fetch_generic(0);
        store(0, 0);
    }
    __syncthreads();
    MZC_T(0);
    // B operands run two pixel tiles ahead of the MFMAs in a 3-slot ring; step n = tap * NPT + pt lives in xr[n % 3]
    float4 xr[3];
    constexpr int FL = WHOLE ? 1 : 2 * CONV_RK;     // staging loads per tap: whole images 1 (taps 0..3), tiled 2 channels (taps 0..7)
    constexpr int FT = WHOLE ? 4 : 8;               // taps that carry staging loads
    constexpr int FS = FL < NPT ? FL : NPT;         // of which this many go one per pixel tile, the rest in front
    for (int cb = 0; cb < n_cb; cb++) {
        const float* sb = slab + (cb & 1) * bufsz;
        xr[0] = *reinterpret_cast<const float4*>(sb + off[0]);
        xr[1] = *reinterpret_cast<const float4*>(sb + (NPT > 1 ? off[NPT > 1 ? 1 : 0] : off[0] + 4));
        __builtin_amdgcn_sched_barrier(0);
        MZC_T(1);
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
#pragma unroll
            for (int c = 0; c < NCT; c++) wr[(tap + WD - 1) % WD][c] = wload(c, cb * 9 + tap + WD - 1);  // weights WD - 1 steps ahead
            if constexpr (WHOLE) {
                if (tap < 4) wfetch_real(cb + 1 < n_cb ? cb + 1 : cb, tap);  // next block's slab, one channel per tap
            } else {
                if (tap < 8) fetch_part(cb + 1 < n_cb ? cb + 1 : cb, 2 * tap, 2 * tap + 2);  // next block's slab, 2 channels per tap
            }
#pragma unroll
            for (int pt = 0; pt < NPT; pt++) {
                const int n = tap * NPT + pt, n2 = n + 2;
                if (MZC_XS_READ && n2 < 9 * NPT) {  // A operand two steps ahead
                    const int tap2 = n2 / NPT, pt2 = n2 - tap2 * NPT;
                    xr[n2 % 3] = *reinterpret_cast<const float4*>(sb + off[pt2] + ((tap2 / 3) * siw + (tap2 % 3)) * 4);
                }
                const float4 x4 = xr[MZC_XS_READ ? n % 3 : 0];
#pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const float4 w4 = wr[tap % WD][c];
                    acc[c][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.x, w4.x, acc[c][pt], 0, 0, 0);
                    acc[c][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.y, w4.y, acc[c][pt], 0, 0, 0);
                    acc[c][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.z, w4.z, acc[c][pt], 0, 0, 0);
                    acc[c][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.w, w4.w, acc[c][pt], 0, 0, 0);
                }
            }
            // schedule: the weight loads (and staging loads that have no pixel tile to hide behind) first, then per pixel tile
            // one LDS read, one staging load, its MFMAs
#ifdef MZC_NO_FETCH
            __builtin_amdgcn_sched_group_barrier(0x020, NCT, 0);
#else
            __builtin_amdgcn_sched_group_barrier(0x020, NCT + FL - FS, 0);  // (tap 8 has no staging loads: the group just comes up short)
#endif
#pragma unroll
            for (int pt = 0; pt < NPT; pt++) {
                if (MZC_XS_READ && tap * NPT + pt + 2 < 9 * NPT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#ifndef MZC_NO_FETCH
                if (tap < FT && pt < FS) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#endif
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * NCT, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cb + 1 < n_cb && !pure_real(cb + 1)) {
            if constexpr (WHOLE) wfix_generic(cb + 1);
            else fetch_generic(cb + 1);
        }
        MZC_T(2);
#ifndef MZC_NO_STORE
        if (cb + 1 < n_cb) {
            if constexpr (WHOLE) wstore((cb + 1) & 1);
            else store((cb + 1) & 1, cb + 1);
        }
#endif
        MZC_T(3);
        __syncthreads();
        MZC_T(4);
    }
    // ---- epilogue: lane (q, j) holds pixel slots pt*16 + 4q + r (r = 0..3) of output channel 16*tile + j.  Slots that are
    // four consecutive pixels of one image row go out as one 16-byte buffer store (residual: one 16-byte load), others one
    // by one; residual values are fetched EC pixel tiles at a time ----
    const int ohw = L.oh * L.ow;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(L.out, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.residual ? L.residual : L.out), 0, -1, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_spw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(SP ? L.sp_w : L.out), 0, -1, 0x00020000);
    constexpr int EC = NPT < 4 ? NPT : 4;
#pragma unroll
    for (int c = 0; c < NCT; c++) {
        if (cot[c] >= co_tiles) continue;  // wave-uniform
        const int co = cot[c] * 16 + j;
        const bool co_ok = co < L.cout;
        const unsigned soff = (unsigned)((size_t)img0 * L.cout * ohw * sizeof(float));
        // Round 6 (found on the learner's conv first, mz_learn_conv.h): the residual loads of a batch sat inside per-lane `if (vec)` regions, so every batch
        // of EC tiles was a dependent round trip to L2 / HBM behind the previous batch's stores.  Whole-image builds: where a batch is regular -- every
        // lane's four slots are one whole quad of one image (pixel count a multiple of four, or the workgroup's only image and the tiles inside it) --
        // the residual comes by unconditional 16-byte loads (padding lanes read a clamped address and discard) and batch it + 1's are issued before
        // batch it is finished and stored.  nfast: the leading regular batches (workgroup-uniform; compile-time in the SIDE builds).
        constexpr int NB = (NPT + EC - 1) / EC;
        int nfast = 0;
        struct Bt { f32x4 rv[EC]; unsigned vo[EC]; bool valid[EC]; };
        Bt bt[2];
        [[maybe_unused]] auto issue = [&](int p0, Bt& b) {
#pragma unroll
            for (int e = 0; e < EC; e++) {
                const int p = (p0 + e) * 16 + 4 * q;
                const int g = conv_idiv(p, r_tp), pp = p - g * TP;
                b.valid[e] = co_ok && (p0 + e < NPT) && (g < L.G) && (img0 + g < L.B);
                const int gs = b.valid[e] ? g : 0, cs = co_ok ? co : 0, ps = b.valid[e] ? pp : 0;  // (a real address for every lane)
                b.vo[e] = (unsigned)((gs * L.cout + cs) * ohw + ps) * (unsigned)sizeof(float);
                b.rv[e] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                if (L.residual && p0 + e < NPT) {
                    const conv_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_res, b.vo[e], soff, 0);
                    b.rv[e] = f32x4{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
                }
            }
        };
        [[maybe_unused]] auto finish = [&](int p0, const Bt& b) {
#pragma unroll
            for (int e = 0; e < EC; e++) {
                if (p0 + e >= NPT) continue;
                f32x4 v = acc[c][p0 + e];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float t = v[r] + (b.valid[e] ? b.rv[e][r] : 0.0f);
                    if (L.relu && !(t > 0.0f)) t = 0.0f;
                    v[r] = t;
                }
                if (b.valid[e])
                    __builtin_amdgcn_raw_buffer_store_b128(conv_u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])},
                                                           rs_out, b.vo[e], soff, 0);
            }
        };
#if !defined(MZC_NO_EPI) && !defined(MZC_NO_EPI_PIPE)
        if constexpr (WHOLE && !SP) {
#pragma unroll
            for (int it = 0; it < NB; it++)
                if (nfast == it && ((ohw & 3) == 0 || (L.G == 1 && (it * EC + EC) * 16 <= ohw))) nfast = it + 1;
            if (nfast > 0) issue(0, bt[0]);
        }
#endif
#pragma unroll
        for (int p0 = 0; p0 < NPT; p0 += EC) {
            if constexpr (WHOLE && !SP) {
                const int it = p0 / EC;
                if (it < nfast) {
                    if (it + 1 < nfast) issue(p0 + EC, bt[(it + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    finish(p0, bt[it & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
            }
            unsigned vo[EC][4];
            bool ok[EC][4], vec[EC];
            f32x4 rv[EC];
            [[maybe_unused]] int tr[EC][4];  // SP: LDS row (12 ints) of the slot's action terms
#pragma unroll
            for (int e = 0; e < EC; e++) {
                const int pt = p0 + e < NPT ? p0 + e : NPT - 1;
                const int p = pt * 16 + 4 * q;
                int g = conv_idiv(p, r_tp), pp = p - g * TP, py = conv_idiv(pp, r_tw), px = pp - py * L.tw;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    ok[e][r] = co_ok && (p0 + e < NPT) && (g < L.G) && (img0 + g < L.B) && (ty0 + py < L.oh) && (tx0 + px < L.ow);
                    vo[e][r] = (unsigned)((g * L.cout + co) * ohw + (ty0 + py) * L.ow + tx0 + px) * (unsigned)sizeof(float);
                    if constexpr (SP) tr[e][r] = ok[e][r] ? (g * ohw + (ty0 + py) * L.ow + tx0 + px) * 12 : 0;
                    px++;
                    if (px == L.tw) { px = 0; py++; }
                    if (py == L.th) { py = 0; g++; }
                }
                vec[e] = ok[e][0] && ok[e][1] && ok[e][2] && ok[e][3] && vo[e][1] == vo[e][0] + 4 && vo[e][2] == vo[e][0] + 8 && vo[e][3] == vo[e][0] + 12;
                rv[e] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#ifndef MZC_NO_EPI
                if (L.residual) {
                    if (vec[e]) {
                        const conv_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_res, vo[e][0], soff, 0);
                        rv[e] = f32x4{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (ok[e][r]) rv[e][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_res, vo[e][r], soff, 0));
                    }
                }
#endif
            }
            // SP: the nine weights of each of the lane's four pixels, requested one pixel tile ahead of the additions
            [[maybe_unused]] float spw[2][4][9];
            [[maybe_unused]] auto sp_request = [&](int e, int buf) {
                if constexpr (SP) {
                    const int* s_term = reinterpret_cast<const int*>(slab + 2 * bufsz);
                    const unsigned cob = (unsigned)(co_ok ? co : 0) * (unsigned)sizeof(float);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int4 t0 = *reinterpret_cast<const int4*>(s_term + tr[e][r]), t1 = *reinterpret_cast<const int4*>(s_term + tr[e][r] + 4),
                                   t2 = *reinterpret_cast<const int4*>(s_term + tr[e][r] + 8);
                        const int ks[9] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x};
#pragma unroll
                        for (int t = 0; t < 9; t++)
                            spw[buf][r][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_spw, (unsigned)ks[t] + cob, 0, 0));
                    }
                }
            };
            if constexpr (SP) sp_request(0, 0);
#pragma unroll
            for (int e = 0; e < EC; e++) {
                if (p0 + e >= NPT) continue;
                if constexpr (SP) {
                    if (e + 1 < EC && p0 + e + 1 < NPT) sp_request(e + 1, (e + 1) & 1);
                }
                f32x4 v = acc[c][p0 + e];
#ifdef MZC_NO_EPI
                if (v[0] != 123.456f) continue;
#endif
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float t = v[r] + rv[e][r];  // (+ 0 without a residual: exact; -0 + 0 = +0 is clamped the same way)
                    if constexpr (SP) {         // k_action_sparse's order: the stored pre-activation, then the nine terms, then the ReLU
#pragma unroll
                        for (int i = 0; i < 9; i++) t = t + spw[e & 1][r][i];
                    }
                    if (L.relu && !(t > 0.0f)) t = 0.0f;
                    v[r] = t;
                }
                if (vec[e]) {
                    __builtin_amdgcn_raw_buffer_store_b128(conv_u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])},
                                                           rs_out, vo[e][0], soff, 0);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (ok[e][r]) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), rs_out, vo[e][r], soff, 0);
                }
            }
        }
    }
    MZC_T(5);
    MZC_T_FLUSH(L);
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

// The dynamics net's action planes, exactly, without the dense work.  network.py:440-444 builds the [A, h, w] block whose flat
// element f = c*h*w + pixel is 1 iff f % A == action.  When gcd(h*w, A) == 1 (every board game: A = h*w + 1) exactly ONE
// action channel is 1 at any pixel: c = (action - pixel) * (h*w)^-1 mod A.  So of the A * 9 action-plane terms of an output
// only <= 9 are non-zero (one per in-image tap), each adding fma(1, w, acc) = acc + w; all the others are fma(0, w, acc) = acc.
// The dense kernel therefore runs over the REAL channels only (they fill the leading 16-channel blocks: planes % 16 == 0) and
// writes the pre-activation; this kernel adds the <= 9 weights in the chain's order -- (16-channel block, tap, channel) -- and
// applies the ReLU: bit-identical to the dense evaluation (the sign of a zero accumulator is the only thing the skipped terms
// could change, and the ReLU stores +0 for both).  C5: 23 channel blocks become 8.
struct ActionSparseLaunch {
    float* x;              // dense [B][cout][hw]: pre-activation in, ReLU(pre-activation + action terms) out
    const int* action;     // [B]
    const float* w;        // folded weights of the action channels [cout][A][9]
    int B, cout, h, w_img, A;
    int inv_hw;            // (h*w)^-1 mod A
};

__global__ __launch_bounds__(256) void k_action_sparse(const ActionSparseLaunch L) {
    __shared__ int s_term[16][9];   // per pixel of the chunk: weight index c*9 + tap of its k-th term in chain order, -1 = none
    const int hw = L.h * L.w_img, b = blockIdx.y, p0 = blockIdx.x * 16, tid = threadIdx.x;
    if (tid < 16) {
        const int p = p0 + tid;
        int key[9];
#pragma unroll
        for (int t = 0; t < 9; t++) {
            key[t] = 0x7fffffff;
            const int y = p / L.w_img + t / 3 - 1, x = p % L.w_img + t % 3 - 1;
            if (p < hw && y >= 0 && y < L.h && x >= 0 && x < L.w_img) {
                const int nb = y * L.w_img + x;
                int d = (L.action[b] - nb) % L.A;
                d = d < 0 ? d + L.A : d;
                const int c = (int)(((long long)d * L.inv_hw) % L.A);
                key[t] = ((c >> 4) * 9 + t) * 16 + (c & 15);  // chain order: block, tap, channel in block
            }
        }
#pragma unroll
        for (int i = 1; i < 9; i++) {  // insertion sort, fully unrolled (registers)
#pragma unroll
            for (int k = i; k > 0; k--) {
                const int a = key[k - 1], c2 = key[k];
                key[k - 1] = a < c2 ? a : c2;
                key[k] = a < c2 ? c2 : a;
            }
        }
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int k = key[t];
            s_term[tid][t] = k == 0x7fffffff ? -1 : (((k / 144) * 16 + (k & 15)) * 9 + (k % 144) / 16);
        }
    }
    __syncthreads();
    const int px = tid & 15, p = p0 + px;
    if (p >= hw) return;
    int term[9];
#pragma unroll
    for (int t = 0; t < 9; t++) term[t] = s_term[px][t];
    for (int co = tid >> 4; co < L.cout; co += 16) {
        float* o = L.x + ((size_t)b * L.cout + co) * hw + p;
        const float* w = L.w + (size_t)co * L.A * 9;
        float acc = *o;
        float wv[9];
#pragma unroll
        for (int t = 0; t < 9; t++) {  // branch-free: all nine loads in flight, absent terms add 0 (acc + 0 == acc up to the sign of
            const int k = term[t];     // a zero, which the ReLU erases)
            const float v = w[k < 0 ? 0 : k];
            wv[t] = k < 0 ? 0.0f : v;
        }
#pragma unroll
        for (int t = 0; t < 9; t++) acc = acc + wv[t];
        *o = acc > 0.0f ? acc : 0.0f;
    }
}

// normalize_hidden_state for conv states (util.py:31-36): min/max over the channels of each pixel (exact, so any reduction
// order gives the reference's result).  in dense [B][C][hw]; written to every non-null destination: out_ptrs[b] (node store
// rows), dense out_a, dense out_b.  Workgroup = one image x 32 pixels; thread (pixel, channel group of 8): its C/8 channels
// stay in registers between the min/max pass and the scaling pass; HBM traffic = one read + one write per destination.
// CPT = channels per thread held in registers (C <= 8 * CPT)
template <int CPT>
__global__ __launch_bounds__(256) void k_normalize_planes(const float* in, float* const* out_ptrs, float* out_a, float* out_b, int B, int C, int hw) {
    __shared__ float s_mn[8][32], s_mx[8][32];
    const int b = blockIdx.y, px = threadIdx.x & 31, cg = threadIdx.x >> 5, p = blockIdx.x * 32 + px;
    const bool ok = p < hw;
    const int cpt = (C + 7) >> 3;  // channels per thread: cg * cpt .. +cpt
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
    float* o0 = out_ptrs ? out_ptrs[b] + p : nullptr;
    float* o1 = out_a ? out_a + (size_t)b * C * hw + p : nullptr;
    float* o2 = out_b ? out_b + (size_t)b * C * hw + p : nullptr;
#pragma unroll
    for (int i = 0; i < CPT; i++) {
        const int c = cg * cpt + i;
        if (i < cpt && c < C) {
            const float r = (v[i] - mn) / d;
            if (o0) o0[(size_t)c * hw] = r;
            if (o1) o1[(size_t)c * hw] = r;
            if (o2) o2[(size_t)c * hw] = r;
        }
    }
}

// Head (network.py:424-430, 472-486): 1x1 conv (C -> oc planes, BN folded) + ReLU + flatten + Linear(oc*hw -> n_out),
// then for value/reward heads softmax-expectation-transform (util.py:70-93) or, for the policy head, softmax.
// One workgroup of 256 threads per image; every dot product is one fmaf chain in the oracle's order (channels ascending,
// then features ascending).  The input is staged through LDS in chunks of HEAD_CK channels by all threads (coalesced,
// independent loads) so that the chains run from LDS instead of paying a global-memory latency per channel.
constexpr int HEAD_CK = 16;

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
    float* stage = lg + ((L.n_out + 3) & ~3);         // [HEAD_CK][hw]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* src = L.in_ptrs ? L.in_ptrs[b] : L.in + (size_t)b * L.C * L.hw;
    const int nf = L.oc * L.hw;
    // 1x1 conv: feature i = (plane o, pixel p); threads stride over features (at most 2 * 240 of them)
    float acc[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int i = tid + 256 * u;
        acc[u] = i < nf ? L.cb[i / L.hw] : 0.0f;
    }
    for (int c0 = 0; c0 < L.C; c0 += HEAD_CK) {
        const int nc = L.C - c0 < HEAD_CK ? L.C - c0 : HEAD_CK;
        __syncthreads();
        for (int i = tid; i < nc * L.hw; i += 256) stage[i] = src[(size_t)c0 * L.hw + i];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int i = tid + 256 * u;
            if (i < nf) {
                const int o = i / L.hw, p = i - o * L.hw;
                const float* w = L.cw + o * L.C + c0;
                for (int c = 0; c < nc; c++) acc[u] = fmaf(stage[c * L.hw + p], w[c], acc[u]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int i = tid + 256 * u;
        if (i < nf) feat[i] = acc[u] > 0.0f ? acc[u] : 0.0f;
    }
    __syncthreads();
    const int K = nf;
    for (int n = tid; n < L.n_out; n += 256) {
        float a = L.lb[n];
        const float* w = L.lw + (size_t)n * K;
        for (int k = 0; k < K; k++) a = fmaf(feat[k], w[k], a);
        lg[n] = a;
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
