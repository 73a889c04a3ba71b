// mz_device.h -- device-side scalar helpers shared by the planner kernels (gfx950).
//
// Built with -ffp-contract=off: every fused multiply-add below is an explicit fmaf()/fma(); plain a*b+c stays
// two roundings.  sqrtf() and '/' are IEEE-correct (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mz {

__device__ __forceinline__ float bits2f(uint32_t u) { return __uint_as_float(u); }

// exp() used by every softmax on the path (policy network.py:72,100; value/reward util.py:88).
// Cephes-style: n = rint(x*log2e), r = x - n*ln2 in two fmaf steps, degree-5 polynomial, scale by 2^n.
// ~1 ulp; deterministic instruction sequence (no libm, no fast-math).
// Branch-free: out-of-range arguments are clamped for the polynomial and the result is selected at the end (the
// same values as early returns; straight-line code can be scheduled into MFMA shadows, branches cannot).
__device__ __forceinline__ float expf_det(float x0) {
    const float x = __builtin_amdgcn_fmed3f(x0, -104.0f, 89.0f);  // == x0 wherever the polynomial's result is used
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    int ni = (int)n;
    const bool lo = ni < -126, hi = ni > 127;
    y = lo ? y * 5.42101086242752217e-20f : y;  // 2^-64
    ni = lo ? ni + 64 : ni;
    y = hi ? y * 2.0f : y;
    ni = hi ? ni - 1 : ni;
    float res = y * bits2f((uint32_t)(ni + 127) << 23);
    asm volatile("" : "+v"(res));  // keeps the polynomial unconditional: hipcc would turn the selects below back into branches around it
    return x0 > 88.5f ? __uint_as_float(0x7f800000u) : (x0 < -103.5f ? 0.0f : res);
}

// signed_parabolic, util.py:25-28 (eps = 1e-3), float32 in the op order of the torch expression.
__device__ __forceinline__ float signed_parabolic(float x) {
    float ax = fabsf(x);
    float t = 1.001f + ax;
    float u = 0.004f * t;
    float v = 1.0f + u;
    float s = sqrtf(v);
    float z = s / 2.0f / 0.001f - 500.0f;
    float sq = z * z;
    float m = sq - 1.0f;
    float sg = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
    return sg * m;
}

// DPP row rotation inside each 16-lane row (no LDS crossbar, one VALU op): lane i receives lane (i + n) mod 16.
template <int N>
__device__ __forceinline__ float row_ror(float v) {
    // (bound_ctrl: a rotation has no invalid source lanes, and with it hipcc does not pre-load the destination with `old` -- one v_mov less
    // per DPP move on a wave whose every instruction is 5.8 cycles of issue, tools/micro/issue.hip)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ double row_ror(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), 0x120 + N, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x120 + N, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Row reduction order (the test oracle restates the same order in its row_reduce16): each of 16 lanes holds the
// sequential partial sum of elements j, j+16, j+32, ...; the partials are combined by a butterfly with strides
// 8, 4, 2, 1.  Every lane of the 16-lane segment returns the same total.
__device__ __forceinline__ float butterfly16(float part) {
    part = part + row_ror<8>(part);
    part = part + row_ror<4>(part);
    part = part + row_ror<2>(part);
    part = part + row_ror<1>(part);
    return part;
}
// max over the 16 lanes of a row, every lane gets it.  One v_max_f32 with a DPP-rotated operand per step (hipcc's own
// rendering of `o = row_ror(v); v = o > v ? o : v` is mov + dpp mov + compare + select with wait states: 6 instructions a
// step on the critical path of every selection and softmax).  Inputs are never NaN here; -inf is fine.
__device__ __forceinline__ float butterfly16_max(float v) {
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(v));
    return v;
}
__device__ __forceinline__ float butterfly16_min(float v) {
    float o;
    o = row_ror<8>(v); v = o < v ? o : v;
    o = row_ror<4>(v); v = o < v ? o : v;
    o = row_ror<2>(v); v = o < v ? o : v;
    o = row_ror<1>(v); v = o < v ? o : v;
    return v;
}

// register-only version for rows of at most 32 values: lane j holds l0 = row[j], l1 = row[j + 16] (has1 says whether
// the second one exists).  Same arithmetic as row_logits_to_scalar.
__device__ __forceinline__ float row2_logits_to_scalar(float l0, float l1, bool has0, bool has1, int S, int j) {
    // straight-line (selects, no branches): everything is computed for both slots and masked
    const float ninf = __uint_as_float(0xff800000u);
    float m = has0 ? l0 : ninf;
    m = (has1 & (l1 > m)) ? l1 : m;
    m = butterfly16_max(m);
    const float x0 = expf_det((has0 ? l0 : m) - m), x1 = expf_det((has1 ? l1 : m) - m);
    const float e0 = has0 ? x0 : 0.0f, e1 = has1 ? x1 : 0.0f;
    float a = 0.0f;
    a = has0 ? a + e0 : a;
    a = has1 ? a + e1 : a;
    const float sum = butterfly16(a);
    const int half = (S - 1) / 2;
    float t = 0.0f;
    const float p0 = e0 / sum, p1 = e1 / sum;
    const float t0 = p0 * (float)(j - half), t1 = p1 * (float)(j + 16 - half);
    t = has0 ? t + t0 : t;
    t = has1 ? t + t1 : t;
    return signed_parabolic(butterfly16(t));
}

// Two rows at once (the reward and the value head of one env on the same 16 lanes): the same arithmetic as two calls of
// row2_logits_to_scalar, written as ONE basic block so that the two latency-bound chains (DPP reductions, exp polynomial,
// IEEE division, sqrt) interleave; behind the callers' per-head `S == 1` branches they ran one after the other.
__device__ __forceinline__ void butterfly16_max2(float& a, float& b) {
    asm volatile("s_nop 0\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void rows2_logits_to_scalars(const float (&l0)[2], const float (&l1)[2], const bool (&has0)[2], const bool (&has1)[2],
                                                        const int (&S)[2], int j, float (&out)[2]) {
    const float ninf = __uint_as_float(0xff800000u);
    float m[2], e0[2], e1[2], a[2], sum[2], t[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        m[r] = has0[r] ? l0[r] : ninf;
        m[r] = (has1[r] & (l1[r] > m[r])) ? l1[r] : m[r];
    }
    butterfly16_max2(m[0], m[1]);
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const float x0 = expf_det((has0[r] ? l0[r] : m[r]) - m[r]), x1 = expf_det((has1[r] ? l1[r] : m[r]) - m[r]);
        e0[r] = has0[r] ? x0 : 0.0f;
        e1[r] = has1[r] ? x1 : 0.0f;
        a[r] = 0.0f;
        a[r] = has0[r] ? a[r] + e0[r] : a[r];
        a[r] = has1[r] ? a[r] + e1[r] : a[r];
    }
#pragma unroll
    for (int r = 0; r < 2; r++) sum[r] = butterfly16(a[r]);
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int half = (S[r] - 1) / 2;
        const float p0 = e0[r] / sum[r], p1 = e1[r] / sum[r];
        const float t0 = p0 * (float)(j - half), t1 = p1 * (float)(j + 16 - half);
        t[r] = 0.0f;
        t[r] = has0[r] ? t[r] + t0 : t[r];
        t[r] = has1[r] ? t[r] + t1 : t[r];
    }
#pragma unroll
    for (int r = 0; r < 2; r++) t[r] = butterfly16(t[r]);
#pragma unroll
    for (int r = 0; r < 2; r++) out[r] = signed_parabolic(t[r]);
}

// logits_to_transformed_expected_value, util.py:70-93: softmax -> E[linspace(-(S-1)/2, (S-1)/2, S)] -> signed_parabolic,
// computed by the 16 lanes of a segment on one row `lg` (LDS, overwritten with the exponentials).  S == 1: identity.
__device__ __forceinline__ float row_logits_to_scalar(float* lg, int S, int j) {
    if (S == 1) return lg[0];
    float m = __uint_as_float(0xff800000u);
    for (int i = j; i < S; i += 16) m = lg[i] > m ? lg[i] : m;
    m = butterfly16_max(m);
    float a = 0.0f;
    for (int i = j; i < S; i += 16) {
        const float e = expf_det(lg[i] - m);
        lg[i] = e;
        a = a + e;
    }
    const float sum = butterfly16(a);
    const int half = (S - 1) / 2;
    float t = 0.0f;
    for (int i = j; i < S; i += 16) {
        const float p = lg[i] / sum;
        const float tt = p * (float)(i - half);
        t = t + tt;
    }
    return signed_parabolic(butterfly16(t));
}

// softmax of one row of n logits (policy, network.py:72,100) by the 16 lanes of a segment; out may alias lg
__device__ __forceinline__ void row_softmax(const float* lg, float* out, int n, int j) {
    float m = __uint_as_float(0xff800000u);
    for (int i = j; i < n; i += 16) m = lg[i] > m ? lg[i] : m;
    m = butterfly16_max(m);
    float a = 0.0f;
    for (int i = j; i < n; i += 16) {
        const float e = expf_det(lg[i] - m);
        out[i] = e;
        a = a + e;
    }
    const float sum = butterfly16(a);
    for (int i = j; i < n; i += 16) out[i] = out[i] / sum;
}

// numpy's np.sum on a contiguous 1-D array (pairwise_sum, numpy/core/src/umath/loops_utils.h.src):
// the reference normalises the masked prior with it (mcts.py:296) and the play policy (mcts.py:279).
__device__ inline double np_sum_f64(const double* a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r = r + a[i];
        return r;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] = r[j] + a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res = res + a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        // one level of recursion is enough for n <= 256 (both halves <= 128)
        double lo, hi;
        {
            const double* b = a;
            int m = n2;
            double r[8];
            for (int j = 0; j < 8; j++) r[j] = b[j];
            int i;
            for (i = 8; i < m - (m % 8); i += 8)
                for (int j = 0; j < 8; j++) r[j] = r[j] + b[i + j];
            lo = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            for (; i < m; i++) lo = lo + b[i];
        }
        {
            const double* b = a + n2;
            int m = n - n2;
            double r[8];
            for (int j = 0; j < 8; j++) r[j] = b[j];
            int i;
            for (i = 8; i < m - (m % 8); i += 8)
                for (int j = 0; j < 8; j++) r[j] = r[j] + b[i + j];
            hi = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            for (; i < m; i++) hi = hi + b[i];
        }
        return lo + hi;
    }
}

__device__ inline float np_sum_f32(const float* a, int n) {
    if (n < 8) {
        float r = 0.0f;
        for (int i = 0; i < n; i++) r = r + a[i];
        return r;
    }
    // n <= 256 supported (two blocks of <= 128)
    int n2 = n;
    const float* b = a;
    float part[2];
    int parts = 1;
    int lens[2] = {n, 0};
    if (n > 128) {
        n2 = n / 2;
        n2 -= n2 % 8;
        lens[0] = n2;
        lens[1] = n - n2;
        parts = 2;
    }
    for (int p = 0; p < parts; p++) {
        int m = lens[p];
        float r[8];
        for (int j = 0; j < 8; j++) r[j] = b[j];
        int i;
        for (i = 8; i < m - (m % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] = r[j] + b[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < m; i++) res = res + b[i];
        part[p] = res;
        b += m;
    }
    return parts == 1 ? part[0] : part[0] + part[1];
}

// ---- Philox4x32-10 counter-based RNG (production-mode randomness; parity mode injects recorded draws) ----
struct Philox {
    uint32_t key[2];
    uint32_t ctr[4];
    __device__ Philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2) {
        key[0] = (uint32_t)seed;
        key[1] = (uint32_t)(seed >> 32);
        ctr[0] = 0; ctr[1] = c0; ctr[2] = c1; ctr[3] = c2;
    }
    __device__ void round4(uint32_t* out) {
        uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
        uint32_t k0 = key[0], k1 = key[1];
        for (int i = 0; i < 10; i++) {
            uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
            uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
            uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
            uint32_t n1 = (uint32_t)p1;
            uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
            uint32_t n3 = (uint32_t)p0;
            c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
        ctr[0] += 1;  // next block of the same stream
    }
    // uniform double in [0,1) with 53 random bits (same construction as numpy's random_sample: (a>>5, b>>6))
    __device__ double uniform() {
        uint32_t r[4];
        round4(r);
        return ((double)(r[0] >> 5) * 67108864.0 + (double)(r[1] >> 6)) / 9007199254740992.0;
    }
};

// standard_gamma(alpha) -- numpy legacy algorithm (alpha < 1: Ahrens-Dieter style rejection as in
// legacy-distributions.c; alpha == 1: exponential; alpha > 1: Marsaglia-Tsang).  Production noise only.
__device__ inline double gamma_sample(Philox& g, double alpha) {
    if (alpha == 1.0) return -log(1.0 - g.uniform());
    if (alpha < 1.0) {
        // float32 hardware log2 / exp2 (v_log_f32, v_exp_f32) with the power's exponent split off into a float64 ldexp: the
        // rejection loop runs until the slowest of a wave's 64 lanes accepts (~5 rounds), and with ocml's float64 log + pow a
        // round was ~1.5 k cycles -- 13 k cycles per move for noise whose distribution, not whose bits, is the contract
        // (tests/test_gpu_rng.py).  The split keeps float64's range: for alpha = 0.03 a plain float32 power underflows for
        // U < 0.07, both components of a 2-action draw vanish and the uniform fallback biases the variance (measured).
        const float ia = (float)(1.0 / alpha), fa = (float)alpha, ln2 = 0.69314718056f;
        auto pow2d = [](float t) {  // 2 ** t as a double, t clamped to the float64 exponent range
            t = __builtin_amdgcn_fmed3f(t, -1070.0f, 1020.0f);
            const float fl = floorf(t);
            return ldexp((double)__builtin_amdgcn_exp2f(t - fl), (int)fl);
        };
        for (int it = 0; it < 256; it++) {
            // (float)uniform() can round up to 1.0f (probability 2^-25 per draw): then 1 - U == 0, Y == +inf and the acceptance
            // test inf <= inf would pass a clamped 2^1020 -- the float64 original cannot reach U == 1, so neither may this
            const float one_m = 0x1.fffffep-1f;  // 1 - 2^-24
            const float U = fminf((float)g.uniform(), one_m);
            const double V = (double)(-ln2 * __builtin_amdgcn_logf(1.0f - fminf((float)g.uniform(), one_m)));
            if (U <= 1.0f - fa) {
                const double X = pow2d(ia * __builtin_amdgcn_logf(U));
                if (X <= V) return X;
            } else {
                const float Y = -ln2 * __builtin_amdgcn_logf((1.0f - U) / fa);
                const double X = pow2d(ia * __builtin_amdgcn_logf(1.0f - fa + fa * Y));
                if (X <= V + (double)Y) return X;
            }
        }
        return 0.0;
    }
    double b = alpha - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * b);
    for (int it = 0; it < 256; it++) {
        double X, V;
        do {
            double u1 = g.uniform(), u2 = g.uniform();
            X = sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586 * u2);
            V = 1.0 + c * X;
        } while (V <= 0.0);
        V = V * V * V;
        double U = g.uniform();
        if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return b * V;
        if (log(U) < 0.5 * X * X + b * (1.0 - V + log(V))) return b * V;
    }
    return b;
}

// v ** e for the play policy (mcts.py:276-277).  v is a visit count and every shipped temperature gives an integer exponent
// (1, 2, 4, 5; 10 for the board games' T = 0.1): square-and-multiply is EXACT while the result stays below 2**53 (every
// intermediate product is an integer below it), hence identical to libm's pow there; anything else goes to ocml pow
// (~2 k cycles per call: at T = 0.1 ten of them per move were 3 % of a TicTacToe move).
__device__ inline double pow_policy(double v, double e) {
    const int n = (int)e;
    if ((double)n == e && n >= 1 && n <= 64 && v >= 0.0 && v == floor(v)) {
        double r = 1.0, b = v;
        for (int k = n; k > 0; k >>= 1) {
            if (k & 1) r *= b;
            b *= b;
        }
        if (r < 9007199254740992.0) return r;
    }
    return pow(v, e);
}

}  // namespace mz
